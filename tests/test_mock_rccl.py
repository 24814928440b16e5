"""Several ranks through the library's RCCL halo path on ONE GPU.  The real RCCL refuses two ranks on a device, so
tests/mock_rccl/libmock_rccl.so stands in for it (WDPM_RCCL_LIB): ncclSend / ncclRecv become event-ordered
device-to-device copies, everything above the wire is the product's code - wdpm_comm_init_all, the rank threads each
issuing their own grouped send/recv on their context's stream, the refresh between iteration groups with the
overlapped last iteration, tile flags and the folded max diff around a refresh, the drain module's scalars."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
MOCK = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")


@pytest.fixture(scope="module")
def mock_env():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "mock_rccl")], stdout=subprocess.DEVNULL)
    return dict(os.environ, WDPM_RCCL_LIB=MOCK, WDPM_HALO="rccl", WDPM_RCCL_SHARED_DEVICE_OK="1")


def test_groups_with_rccl_halos_equal_one_context(mock_env):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mock_rccl_worker.py")], cwd=ROOT, env=mock_env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "MOCK_RCCL_GROUPS_OK 5" in p.stdout, p.stdout + p.stderr[-2000:]
    assert "MOCK_RCCL_RANDOM_JOBS_OK" in p.stdout, p.stdout + p.stderr[-2000:]


def test_cli_on_three_slabs_with_rccl_halos(mock_env, tmp_path):
    """the shipped WDPMCL binary, WDPM_DEVICES=0,0,0, halos by (stand-in) RCCL: the validation chain's first step
    stays byte-identical to the reference's report and raster"""
    import gzip
    import hashlib
    import json
    from conftest import GOLDEN
    from test_cli import HIP_CLI, file_sha, strip_timing
    golden = json.load(open(os.path.join(GOLDEN, "basin5_cli.json")))
    with gzip.open(os.path.join(GOLDEN, "basin5.asc.gz"), "rb") as f:
        (tmp_path / "basin5.asc").write_bytes(f.read())
    g = golden["cfg2_add300_k1000"]
    env = dict(mock_env, WDPM_DEVICES="0,0,0", WDPM_EXCHANGE_EVERY="3")
    p = subprocess.run([HIP_CLI] + g["args"], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr
    assert "halos by RCCL send/recv" in p.stderr
    assert hashlib.sha256(strip_timing(p.stdout).encode()).hexdigest() == g["report_sha256_nontiming"]
    assert file_sha(os.path.join(tmp_path, "a300.asc")) == g["out_sha256"]

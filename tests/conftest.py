import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libwdpm_oracle.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/wdpm_oracle.c), built on demand.  The checker, never the product."""
    import wdpm_amd
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return wdpm_amd.load(ORACLE_SO)


@pytest.fixture(scope="session")
def hip():
    """The product library; fails loudly (no fallback) if it has not been built."""
    import wdpm_amd
    return wdpm_amd.load_hip()


@pytest.fixture(scope="session")
def stencil_cases():
    z = np.load(os.path.join(GOLDEN, "stencil_cases.npz"))
    index = json.loads(bytes(z["index_json"]).decode())
    return z, index


@pytest.fixture(scope="session")
def basin5():
    import gzip
    with gzip.open(os.path.join(GOLDEN, "basin5.asc.gz"), "rt") as f:
        hdr = [f.readline().split() for _ in range(6)]
        vals = np.array(f.read().split(), dtype=np.float64)
    ncols, nrows = int(float(hdr[0][1])), int(float(hdr[1][1]))
    return vals.reshape(nrows, ncols), {h[0]: float(h[1]) for h in hdr}

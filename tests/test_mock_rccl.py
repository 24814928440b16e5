"""Several ranks through the library's RCCL halo path on ONE GPU.  The real RCCL refuses two ranks on a device, so
tests/mock_rccl/libmock_rccl.so stands in for it (WDPM_RCCL_LIB): ncclSend / ncclRecv become event-ordered
device-to-device copies, everything above the wire is the product's code - wdpm_comm_init_all, the rank threads each
issuing their own grouped send/recv on their context's stream, the refresh between iteration groups with the
overlapped last iteration, tile flags and the folded max diff around a refresh, the drain module's scalars.
Second half: the same with ONE PROCESS PER RANK, the way bench.py's ranks and the driver's 8-GPU run go -
wdpm_comm_unique_id -> broadcast -> ncclCommInitRank -> send/recv through IPC-mapped device memory -> ncclAllGather -
including bench.py itself on two and eight ranks, and its way out when set-up or the first transfer never comes back."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT
from helpers import bits_equal, n_bit_diff

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


def test_a_rank_that_fails_between_collectives_ends_the_run(mock_env, tmp_path):
    """one rank of a thread group (WDPMCL with WDPM_DEVICES=0,0,0) fails at its third halo transfer.  The others
    have matching receives queued whose sender will never post: the failing rank ends every communicator (wdpm_comm_abort),
    their waits return, and the command exits non-zero with the reason - within seconds, where it used to hang for ever"""
    import gzip
    import time
    from conftest import GOLDEN
    from test_cli import HIP_CLI
    with gzip.open(os.path.join(GOLDEN, "basin5.asc.gz"), "rb") as f:
        (tmp_path / "basin5.asc").write_bytes(f.read())
    env = dict(mock_env, WDPM_DEVICES="0,0,0", WDPM_EXCHANGE_EVERY="3", MOCK_RCCL_FAIL_RANK="1", MOCK_RCCL_FAIL_AFTER="2",
               WDPM_SYNC_TIMEOUT_S="20")
    t = time.time()
    p = subprocess.run([HIP_CLI, "add", "basin5.asc", "NULL", "out.asc", "NULL", "100", "1.0", "1.0", "1", "1", "0.005", "2000"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode not in (0, 42), p.stdout[-500:] + p.stderr[-1500:]
    assert "mock RCCL error" in p.stderr or "aborted" in p.stderr, p.stderr[-1500:]
    assert time.time() - t < 60 and not os.path.exists(tmp_path / "out.asc")


def test_cli_ends_when_the_first_transfer_never_comes_back(mock_env, tmp_path):
    """the DEADLINE path of the shipped binary.  WDPMCL on three slabs (one host thread per rank); rank 1's first
    halo transfer does not return (a fabric that never answers: the stand-in holds the call for a minute).  The library's
    deadline for a communicator's first transfer (WDPM_RCCL_TIMEOUT_S) ends that wait, the rank's failure ends every rank's
    communicator, and the command exits non-zero with the reason on stderr and WITHOUT an output raster - the reference
    exits on any device error (WDPMCL.c:92-118,225-232) - well before the stalled call would have come back"""
    import gzip
    import time
    from conftest import GOLDEN
    from test_cli import HIP_CLI
    with gzip.open(os.path.join(GOLDEN, "basin5.asc.gz"), "rb") as f:
        (tmp_path / "basin5.asc").write_bytes(f.read())
    env = dict(mock_env, WDPM_DEVICES="0,0,0", WDPM_EXCHANGE_EVERY="3", MOCK_RCCL_BLOCK_FIRST_S="60", MOCK_RCCL_BLOCK_RANK="1",
               WDPM_RCCL_TIMEOUT_S="5", WDPM_SYNC_TIMEOUT_S="20")
    t = time.time()
    p = subprocess.run([HIP_CLI, "add", "basin5.asc", "NULL", "out.asc", "NULL", "100", "1.0", "1.0", "1", "1", "0.005", "2000"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=120, env=env)
    took = time.time() - t
    assert p.returncode not in (0, 42), p.stdout[-500:] + p.stderr[-1500:]
    assert "the first RCCL halo transfer did not return within 5 s" in p.stderr, p.stderr[-1500:]
    assert took < 45, f"{took:.0f} s: the command waited for the stalled call"
    assert not os.path.exists(tmp_path / "out.asc")


# ---- one process per rank ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def proc_env(mock_env):
    return dict(mock_env, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))


@pytest.mark.parametrize("module,R,C,world,k,blocks,ponds", [
    ("add", 300, 420, 2, 3, [13, 8], None),
    ("drain", 330, 250, 3, 1, [7, 5], None),
    ("drain", 240, 380, 2, 2, [9, 6], None),
    # big enough for the marching kernel, the overlapped last iteration, dry tiles and the folded max diff
    ("add", 1300, 1100, 3, 4, [21, 10], [[100, 160, 300, 420, 0.3], [900, 960, 100, 200, 0.2]]),
])
def test_rank_processes_with_rccl_halos_equal_one_context(oracle, hip, proc_env, module, R, C, world, k, blocks, ponds):
    """wdpm_rank_create with an id in `world` PROCESSES sharing the GPU: owned rows, every block's max diff and the drain
    module's scalars (through ncclAllGather, rank-chained volume sum included) equal one context's"""
    from test_rowblock import run_ranks, single
    case = dict(seed=70 + world, R=R, C=C, module=module, k=k, thres=1e-4, blocks=blocks, halo="rccl", ctx_kw=dict(device=0))
    if ponds:
        case["ponds"] = ponds
    ref = dict(case, ctx_kw={})
    want, mds = single(oracle, ref)
    got = run_ranks(world, case, libpath=hip.path, env=proc_env)
    for g in got:
        lo, hi = int(g["lo"]), int(g["hi"])
        assert bits_equal(g["own"], want[lo:hi + 1]), f"rows {lo}..{hi}: {n_bit_diff(g['own'], want[lo:hi + 1])} cells differ"
        assert list(g["mds"]) == mds
        if module == "drain":
            assert g["stats"].tolist() == ref["_stats"]


def test_the_multi_gpu_tests_rehearsed_on_one_gpu(proc_env):
    """tests/test_multi_gpu.py lights up by itself on a box with two or more GPUs - which this build has never had.  So that the
    first such lease does not trip over the tests themselves, they run here in a child process that believes in two devices
    (WDPM_TEST_FAKE_NDEV=2, both mapped onto this GPU) with the stand-in RCCL bound: two ranks through wdpm_comm_init_all and as
    rank processes, BASELINE configs 4 and 5 at full size on two row blocks over the library's RCCL path against the REFERENCE's
    bits, bench.py --gpus 2 at 16384^2 with the per-rank attribution.  Everything but the wire."""
    env = dict(proc_env, WDPM_TEST_FAKE_NDEV="2")
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_multi_gpu.py", "-m", "gpu", "-x", "-q", "-rs", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1700)
    tail = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
    assert p.returncode == 0 and " passed" in tail and "skipped" not in tail and "failed" not in tail, p.stdout[-4000:] + p.stderr[-2000:]


def bench_line(env, *args, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--no-cpu-baseline"], capture_output=True,
                       text=True, timeout=timeout, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout + p.stderr[-2000:]
    return json.loads(lines[0]), p.stderr


@pytest.fixture(scope="module")
def one_rank_line():
    return bench_line(dict(os.environ), "--size", "2048", "--steps", "13", "--warmup", "3")[0]


def test_bench_ranks_over_mock_rccl_processes(proc_env, one_rank_line):
    """bench.py --gpus 2 as the driver's launcher runs it (torch.distributed.run, one process per rank), halos by the
    library's RCCL path over the stand-in: gloo control plane, communicator from the broadcast id, the rehearsal block,
    K timed steps ending in ncclAllGather - and the same bits as one rank"""
    d, err = bench_line(proc_env, "--gpus", "2", "--size", "2048", "--steps", "13", "--warmup", "3")
    c = d["config"]
    assert d["n_gpus"] == 2 and c["halo"] == "rccl" and c["rccl_ranks"] == 2 and c["dist_backend"] == "gloo", c
    assert "degraded" not in d and "2.99.99" in c["rccl"] and c["exchange_every"] == 8
    # ... and cannot be mistaken for a two-GPU measurement: the line names the GPU behind every rank and the library
    # the RCCL entry points came from, and calls itself a rehearsal for both reasons
    assert d["rehearsal"] is True and c["distinct_gpus"] == 1 and len(c["devices"]) == 2, d
    assert c["devices"][0]["pci_bus_id"] == c["devices"][1]["pci_bus_id"] and c["devices"][0]["pid"] != c["devices"][1]["pid"]
    assert "libmock_rccl" in c["rccl"] and "stand-in" in d["rehearsal_reason"] and "2 ranks on 1 physical GPU" in d["rehearsal_reason"]
    assert "rehearsal" not in one_rank_line and one_rank_line["config"]["distinct_gpus"] == 1
    assert c["max_diff_m"] == one_rank_line["config"]["max_diff_m"]
    assert "sends over" in err                              # the stand-in says which wire ran (IPC-mapped memory or a file)
    wire = sorted({ln.split("sends over ")[1] for ln in err.splitlines() if "sends over " in ln})
    print("stand-in RCCL wire between rank processes:", wire)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "mock_rccl_wire.txt"), "w") as f:
        f.write("\n".join(wire) + "\n")


def test_bench_four_ranks_over_mock_rccl_processes(proc_env):
    """the driver's N = 8 command shape on the one GPU of this box (4 ranks: the pool allows six processes on the GPU at
    once, and the test runner and the launcher count)"""
    d4, _ = bench_line(proc_env, "--gpus", "4", "--size", "4096", "--steps", "20", "--warmup", "5")
    d1, _ = bench_line(dict(os.environ), "--size", "4096", "--steps", "20", "--warmup", "5")
    assert d4["n_gpus"] == 4 and d4["config"]["halo"] == "rccl" and d4["config"]["rccl_ranks"] == 4 and "degraded" not in d4
    assert d4["config"]["max_diff_m"] == d1["config"]["max_diff_m"]
    assert d4["rehearsal"] is True and d4["config"]["distinct_gpus"] == 1


def test_the_drivers_scale_command_at_full_size_over_mock_rccl_processes(proc_env):
    """the driver's SCALE command - `bench.py --gpus N --steps 20 --warmup 5` at 16384^2, rank processes under
    torch.distributed.run, the DEM shared through /dev/shm, gloo rendezvous, N-way ncclCommInitRank, the launcher's first time
    limit - with FOUR rank processes: this pool's process guard kills a call with more than six processes on the card, this
    test runner is one of them, and a margin of one is kept (eight rank processes of the row-block driver run on the CPU in
    tests/test_rowblock.py::test_eight_rank_processes_over_gloo).  Same max diff as the one-GPU line of the metric's own
    configuration, inside the launcher's first limit, and marked as the rehearsal it is."""
    t = time.monotonic()
    d, err = bench_line(proc_env, "--gpus", "4", "--steps", "20", "--warmup", "5", timeout=900)
    secs = time.monotonic() - t
    c = d["config"]
    assert d["n_gpus"] == 4 and c["halo"] == "rccl" and c["rccl_ranks"] == 4 and "degraded" not in d, d
    assert c["max_diff_m"] == 0.07270028139273618                     # = one GPU's after 25 iterations (tests/test_multi_gpu.py)
    assert d["rehearsal"] is True and c["distinct_gpus"] == 1 and len({v["pid"] for v in c["devices"]}) == 4
    assert len(d["per_rank"]["kernel_ms_per_iteration"]) == 4 and min(d["per_rank"]["refreshes"]) >= 2
    assert "starting them once more" not in err and secs < 200.0, (secs, err[-2000:])     # self_launch's first limit for this command
    with open(os.path.join(ROOT, "gpurun_out", "bench_gpus4_16384_rehearsal_one_gpu.json"), "w") as f:
        json.dump(dict(d, wall_seconds_of_the_whole_command=round(secs, 1)), f)


def test_bench_goes_on_with_host_halos_when_communicator_setup_hangs(proc_env, one_rank_line):
    """rank 1's ncclCommInitRank does not come back: the library's deadline ends the wait, ALL ranks agree and finish on
    host-staged halos in the same processes - a line marked degraded, the same bits, well inside the driver's 600 s"""
    env = dict(proc_env, MOCK_RCCL_HANG_INIT_RANK="1", MOCK_RCCL_HANG_S="25", WDPM_RCCL_TIMEOUT_S="4", MOCK_RCCL_TIMEOUT_S="8")
    d, err = bench_line(env, "--gpus", "2", "--size", "2048", "--steps", "13", "--warmup", "3", timeout=300)
    assert d["degraded"] is True and "did not return within" in d["degraded_reason"], d
    assert d["config"]["halo"].startswith("host") and d["config"]["rccl_ranks"] is None
    assert d["config"]["max_diff_m"] == one_rank_line["config"]["max_diff_m"]
    assert "every rank switches to host-staged halos" in err


def test_bench_goes_on_with_host_halos_when_the_first_transfer_never_completes(proc_env, one_rank_line):
    """the first halo transfer is queued and the stream never moves on (for 12 s here): the stream wait's deadline ends
    it, the communicator is aborted, all ranks finish on host-staged halos"""
    env = dict(proc_env, MOCK_RCCL_STALL_RECV_S="12", WDPM_SYNC_TIMEOUT_S="3")
    d, err = bench_line(env, "--gpus", "2", "--size", "2048", "--steps", "13", "--warmup", "3", timeout=300)
    assert d["degraded"] is True and "did not finish within" in d["degraded_reason"], d
    assert d["config"]["halo"].startswith("host")
    assert d["config"]["max_diff_m"] == one_rank_line["config"]["max_diff_m"]

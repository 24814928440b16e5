"""Tests that light up BY THEMSELVES on a box with two or more GPUs.

Three rounds of this build have only ever seen one-GPU boxes: every multi-slab GPU test names device 0 several times, the real
RCCL has only run with one rank (tests/test_rccl_transport.py), and the driver's 8-GPU bench has been skipped every round.  The
tests here take their device list from torch.cuda.device_count() (wdpm_amd.rowblock.spread_over_devices) and are skipped with a
reason on one GPU; on the first multi-GPU lease they push halo rows over xGMI with the real RCCL and hold the result against the
REFERENCE's bits:

  (a) the library's RCCL path with two real ranks on two devices - ncclCommInitAll (one process) and ncclCommInitRank (one
      process per rank, the way bench.py and the driver's launcher run) - replaces the reference's device set-up, WDPMCL.c:80-121,598-638;
  (b) BASELINE configs 4 and 5 at full size (tests/golden/full_size.npz: 16384^2 add x9, 8192^2 add x3 + drain x9) on one slab per
      device, WDPM_HALO=rccl, default exchange interval, against the reference's sha256 / row hashes / scalars;
  (c) bench.py --gpus <all> as the driver runs it: rccl_ranks == N, no `degraded`, max_diff equal to one rank's, and the per-rank
      attribution (kernel ms, halo-refresh us) present in the line.

The CPU part (no marker) checks the device-list builder."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import wdpm_amd
from conftest import GOLDEN, ROOT
from helpers import bits_equal, n_bit_diff
from wdpm_amd.rowblock import spread_over_devices


def ndev():
    """GPUs on this box; counting them does not initialise any (a process that has must never exec another program)"""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:   # noqa: BLE001 - no torch, no GPU: zero
        return 0


REAL_NDEV = ndev()
# Rehearsal on a one-GPU box (tests/test_mock_rccl.py::test_the_multi_gpu_tests_rehearsed_on_one_gpu): WDPM_TEST_FAKE_NDEV=2 makes these
# tests believe in two devices and maps both onto the one there is; with the stand-in RCCL bound (WDPM_RCCL_LIB) everything but the
# wire - and the claim that more GPUs are faster - is then exercised, so that the first real multi-GPU lease does not trip over the
# tests themselves.
FAKE_NDEV = int(os.environ.get("WDPM_TEST_FAKE_NDEV", "0"))
NDEV = FAKE_NDEV or REAL_NDEV
multi = pytest.mark.skipif(NDEV < 2, reason=f"{NDEV} GPU: needs two or more (runs by itself on a multi-GPU lease)")


def phys(devices):
    return [d % max(REAL_NDEV, 1) for d in devices] if FAKE_NDEV else list(devices)


def test_device_list_builder():
    assert spread_over_devices(1) == [0]
    assert spread_over_devices(8) == list(range(8))
    assert spread_over_devices(16) == list(range(8))                     # a row-block job takes at most eight
    assert spread_over_devices(4, limit=2) == [0, 1]
    assert spread_over_devices(2, 8) == [0, 0, 0, 0, 1, 1, 1, 1]         # neighbouring slabs share a device
    assert spread_over_devices(2, 3) == [0, 0, 1]
    assert spread_over_devices(8, 3) == [0, 1, 2]
    assert spread_over_devices(1, 5) == [0] * 5
    assert spread_over_devices(3, 8) == [0, 0, 0, 1, 1, 1, 2, 2]
    for nd in range(1, 10):
        for ns in range(1, 12):
            d = spread_over_devices(nd, ns)
            assert len(d) == ns and d == sorted(d) and d[0] == 0 and d[-1] == min(nd, 8, ns) - 1
            assert set(d) == set(range(d[-1] + 1))                       # every device up to the last one is used
    with pytest.raises(ValueError):
        spread_over_devices(0)
    with pytest.raises(ValueError):
        spread_over_devices(2, 0)


@pytest.mark.gpu
def test_this_box_reports_its_devices():
    """what the multi-GPU tests below will do on this box: written to gpurun_out/ so that a lease's record says so"""
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "multi_gpu_tests.txt"), "w") as f:
        f.write(f"torch.cuda.device_count() = {REAL_NDEV}{' (rehearsal: pretending ' + str(FAKE_NDEV) + ')' if FAKE_NDEV else ''}: the multi-GPU tests "
                f"{'run on devices ' + str(phys(spread_over_devices(NDEV))) if NDEV >= 2 else 'are skipped'}\n")


# ------------------------------------------------------------------------------------------- (a) real RCCL, two ranks
R, Cc, MISS = 96, 420, -99999.0


def _inputs():
    rng = np.random.default_rng(4)
    bd = np.full((R + 2, Cc + 2), MISS)
    bd[1:-1, 1:-1] = np.round(500 + rng.random((R, Cc)), 3)
    return bd, rng.random((R + 2, Cc + 2)) * (bd > MISS)


@pytest.mark.gpu
@multi
def test_two_real_rccl_ranks_in_one_process(hip):
    """ncclCommInitAll over devices 0 and 1; between kernels, rows 10..14 of each context go to rows 60..64 of the other by
    grouped ncclSend / ncclRecv on the contexts' own streams (two host threads: a grouped send/recv pair blocks until both
    sides have posted), then all-gather; expected: the same moves through the host"""
    import threading
    from wdpm_amd.capi import HaloOp
    assert hip.dll.wdpm_comm_available() == 1, "RCCL could not be bound on a GPU box"
    bd, bw = _inputs()
    bw2 = bw[::-1].copy()
    kw = dict(module="add", nrows=R, ncols=Cc, missingvalue=MISS)
    want = []
    with hip.context(device=0, **kw) as a, hip.context(device=0, **kw) as b:
        a.upload(bd, bw)
        b.upload(bd, bw2)
        a.iterate(3)
        b.iterate(3)
        wa, wb = a.download_water(), b.download_water()
        wa2, wb2 = wa.copy(), wb.copy()
        wa2[60:65], wb2[60:65] = wb[10:15], wa[10:15]
        a.upload_water(wa2)
        b.upload_water(wb2)
        a.iterate(2)
        b.iterate(2)
        want = [a.download_water(), b.download_water()]
    d0, d1 = phys([0, 1])
    with hip.context(device=d0, **kw) as a, hip.context(device=d1, **kw) as b:
        arr = (C.c_void_p * 2)(a._h, b._h)
        hip.check(hip.dll.wdpm_comm_init_all(arr, 2))
        for i, c in enumerate((a, b)):
            n, r = C.c_int32(-1), C.c_int32(-1)
            hip.check(hip.dll.wdpm_comm_size(c._h, C.byref(n), C.byref(r)))
            assert (n.value, r.value) == (2, i)
        a.upload(bd, bw)
        b.upload(bd, bw2)
        got, gathered, errors = [None, None], [None, None], []

        def rank(i, c):
            try:
                c.iterate(3)
                send, recv = (HaloOp * 1)(HaloOp(1 - i, 10, 5)), (HaloOp * 1)(HaloOp(1 - i, 60, 5))
                hip.check(hip.dll.wdpm_comm_exchange(c._h, 1, send, 1, recv))
                c.iterate(2)
                got[i] = c.download_water()
                mine, out = (C.c_double * 2)(1.5 + i, -2.0 * i), (C.c_double * 4)()
                hip.check(hip.dll.wdpm_comm_allgather(c._h, mine, 2, out))
                gathered[i] = list(out)
            except Exception as e:   # noqa: BLE001 - reported below, on the main thread
                errors.append(e)

        th = [threading.Thread(target=rank, args=(i, c)) for i, c in enumerate((a, b))]
        for t in th:
            t.start()
        for t in th:
            t.join(300)
        assert not errors, errors
        assert not any(t.is_alive() for t in th), "a rank thread is still inside RCCL"
        for i in range(2):
            assert bits_equal(got[i], want[i]), f"rank {i}: {n_bit_diff(got[i], want[i])} cells differ after the RCCL transfer"
            assert gathered[i] == [1.5, 0.0, 2.5, -2.0]


@pytest.mark.gpu
@multi
@pytest.mark.parametrize("module,k,blocks", [("add", 4, [21, 10]), ("drain", 2, [9, 6])])
def test_real_rccl_rank_processes_equal_one_context(oracle, hip, module, k, blocks):
    """one PROCESS per GPU (wdpm_comm_unique_id -> broadcast -> ncclCommInitRank -> send/recv -> ncclAllGather), the way bench.py's
    ranks and the driver's launcher go: owned rows, every block's max diff and the drain scalars equal one context's"""
    from test_rowblock import run_ranks, single
    world = len(spread_over_devices(NDEV, limit=4))
    case = dict(seed=90 + world, R=1300, C=1100, module=module, k=k, thres=1e-4, blocks=blocks, halo="rccl",
                devices=phys(spread_over_devices(NDEV, limit=4)), ponds=[[100, 160, 300, 420, 0.3], [900, 960, 100, 200, 0.2]])
    ref = dict(case)
    want, mds = single(oracle, ref)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    got = run_ranks(world, case, libpath=hip.path, env=env)
    for g in got:
        lo, hi = int(g["lo"]), int(g["hi"])
        assert bits_equal(g["own"], want[lo:hi + 1]), f"rows {lo}..{hi}: {n_bit_diff(g['own'], want[lo:hi + 1])} cells differ"
        assert list(g["mds"]) == mds
        if module == "drain":
            assert g["stats"].tolist() == ref["_stats"]


# ------------------------------------------------------------------------------------------- (b) the full-size goldens over xGMI
@pytest.fixture(scope="module")
def golden():
    z = np.load(os.path.join(GOLDEN, "full_size.npz"))
    return z, {m["name"]: m for m in json.loads(bytes(z["index_json"]).decode())}


@pytest.mark.gpu
@multi
def test_config_4_on_one_slab_per_gpu_equals_the_reference(hip, golden, monkeypatch):
    """BASELINE config 4: 16384^2 add, nine iterations at the default exchange interval, one row block per GPU, halos by the real
    RCCL over xGMI - against the reference's bits (WDPMCL.c:1094-1106)"""
    from test_full_size_golden import MISS as M, THRES, assert_matches, inputs
    from wdpm_amd.rowblock import Group
    monkeypatch.setenv("WDPM_HALO", "rccl")
    z, idx = golden
    meta = idx["cfg4_add_16384_i9"]
    n, devices = meta["n"], phys(spread_over_devices(NDEV))
    bd, bw = inputs(hip, n)
    with Group(hip, "add", n, n, M, devices) as g:
        assert g.size == len(devices) and wdpm_amd.HALO_NAMES[g.halo_kind] == "rccl", (g.size, g.halo_kind)
        g.upload(bd, bw)
        md = g.run_block(meta["add_iters"], THRES)
        w = g.download_water()
    assert md == meta["max_diff"]
    assert_matches(z, meta, w)


@pytest.mark.gpu
@multi
def test_config_5_on_one_slab_per_gpu_equals_the_reference(hip, golden, monkeypatch):
    """BASELINE config 5: 8192^2, add x3 then drain x9 on one row block per GPU over the real RCCL: water, max diff, totaldrain,
    |d totaldrain| and the rank-chained sequential volume sum equal the reference's (WDPMCL.c:1076-1093,1257-1268)"""
    from test_full_size_golden import drain_job
    monkeypatch.setenv("WDPM_HALO", "rccl")
    z, idx = golden
    halo = drain_job(hip, z, idx["cfg5_drain_8192_a3_d9"], phys(spread_over_devices(NDEV)), None)
    assert wdpm_amd.HALO_NAMES[halo] == "rccl"


@pytest.mark.gpu
@multi
def test_the_settled_configurations_on_one_slab_per_gpu_equal_the_reference(hip, monkeypatch):
    """round 5: the configurations as SURVEY 8d words them (tests/test_settled_golden.py) on one row block per GPU over the real
    RCCL - config 3's two blocks of 1000 with the flush between them, config 4 after 100 iterations, config 5's drain block from the
    settled add state, and the 12 m ponds on a mostly dry raster (dry tiles and water arriving through refreshed halos)"""
    import test_settled_golden as sg
    monkeypatch.setenv("WDPM_HALO", "rccl")
    z, idx = sg.load_golden()
    devices = phys(spread_over_devices(NDEV))
    done = []
    if "cfg3_add_4096_b2_i2000" in idx:
        sg.job_two_blocks(hip, z, idx, "cfg3_add_4096_i1000", "cfg3_add_4096_b2_i2000", devices)
        done.append("cfg3")
    if "cfg4_add_16384_i100" in idx:
        sg.job_one_block(hip, z, idx["cfg4_add_16384_i100"], devices)
        done.append("cfg4")
    if "cfg5_drain_8192_a1000_d100" in idx:
        upto = [k for k in (100, 1000) if f"cfg5_drain_8192_a1000_d{k}" in idx]
        assert wdpm_amd.HALO_NAMES[sg.job_config5(hip, z, idx, devices, upto=upto)] == "rccl"
        done.append("cfg5")
    if "cfg3x_ponds_4096_b2_i400" in idx:
        sg.job_two_blocks(hip, z, idx, "cfg3x_ponds_4096_i200", "cfg3x_ponds_4096_b2_i400", devices, water=sg.ponds)
        done.append("cfg3x")
    assert done, "no settled entry in tests/golden/full_size.npz"


# ------------------------------------------------------------------------------------------- (b') the shipped binary on every GPU
@pytest.mark.gpu
@multi
def test_the_shipped_wdpmcl_on_every_gpu_of_the_box(hip, tmp_path):
    """`WDPM_GPUS=<all> WDPMCL add ...` - the path a user takes: one host thread per GPU, ncclCommInitAll from those
    threads, halos by RCCL (replaces the reference's pick of ONE OpenCL device, WDPMCL.c:80-121,598-638) - on a synthetic
    2048^2 DEM, two blocks of 1000 iterations: the report (minus wall clock) and the output raster are byte-identical to the
    same command on one GPU."""
    import hashlib
    from test_cli import HIP_CLI, file_sha, strip_timing
    n = 2048
    dem = hip.synth_dem(n, n)
    with open(tmp_path / "dem.asc", "w") as f:
        f.write(f"ncols {n}\nnrows {n}\nxllcorner 0\nyllcorner 0\ncellsize 10\nNODATA_value -99999\n")
        np.savetxt(f, dem, fmt="%.4f")
    devs = phys(spread_over_devices(NDEV))
    runs = {}
    for tag, extra in (("one", {}), ("all", {"WDPM_DEVICES": ",".join(map(str, devs))} if FAKE_NDEV else {"WDPM_GPUS": str(len(devs))})):
        env = {k: v for k, v in os.environ.items() if k not in ("WDPM_GPUS", "WDPM_DEVICES")}
        os.makedirs(tmp_path / tag)                        # (the report names its files: the same names in two directories)
        p = subprocess.run([HIP_CLI, "add", "../dem.asc", "NULL", "out.asc", "NULL", "100", "1.0", "1.0", "1", "1", "0.005", "2000"],
                           cwd=tmp_path / tag, capture_output=True, text=True, timeout=900, env=dict(env, **extra))
        assert p.returncode == 0, p.stderr[-3000:]
        runs[tag] = (hashlib.sha256(strip_timing(p.stdout).encode()).hexdigest(), file_sha(os.path.join(tmp_path, tag, "out.asc")), p.stderr)
    assert runs["all"][0] == runs["one"][0] and runs["all"][1] == runs["one"][1]
    assert "halos by RCCL send/recv" in runs["all"][2] and f"{len(devs)} row blocks" in runs["all"][2], runs["all"][2][-1500:]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"wdpmcl_gpus{len(devs)}.txt"), "w") as f:
        f.write(runs["all"][2])


# ------------------------------------------------------------------------------------------- (c) the bench line of an N-GPU run
def _bench(*args, timeout=900):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout + p.stderr[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@multi
def test_bench_on_every_gpu_of_the_box():
    """the driver's command, --gpus <all> --steps 20 --warmup 5 at 16384^2: every rank on RCCL, nothing degraded, the same max diff
    as one GPU (0.0727... m after 25 iterations), and a line that can explain its own scaling: per rank the kernel's ms per
    iteration and the halo refreshes' count and us"""
    n = len(spread_over_devices(NDEV))
    d = _bench("--gpus", str(n), "--steps", "20", "--warmup", "5")
    one = _bench("--steps", "20", "--warmup", "5")
    c = d["config"]
    assert d["n_gpus"] == n and c["rccl_ranks"] == n and c["halo"] == "rccl", c
    assert "degraded" not in d, d.get("degraded_reason")
    assert c["max_diff_m"] == one["config"]["max_diff_m"] == 0.07270028139273618
    pr = d["per_rank"]
    assert len(pr["kernel_ms_per_iteration"]) == n and all(v > 0 for v in pr["kernel_ms_per_iteration"])
    assert len(pr["refresh_us"]) == n and len(pr["refreshes"]) == n and min(pr["refreshes"]) >= 2
    # the line certifies itself: one PCI bus id per rank, as the library reports them for the ranks' contexts
    devs = c["devices"]
    assert [v["rank"] for v in devs] == list(range(n)) and len({v["pid"] for v in devs}) == n
    if FAKE_NDEV:
        assert d["rehearsal"] is True and c["distinct_gpus"] == 1 and "1 physical GPU" in d["rehearsal_reason"], d
    else:
        assert c["distinct_gpus"] == n == len({v["pci_bus_id"] for v in devs}) and "rehearsal" not in d, (c["devices"], d.get("rehearsal_reason"))
        assert "/librccl" in c["rccl"]               # the ROCm library answered, and says from where
        assert d["value"] > one["value"]             # more GPUs, more cell-updates per second: the least a scaling curve owes
    assert "rehearsal" not in one and one["config"]["distinct_gpus"] == 1
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"bench_gpus{n}.json"), "w") as f:
        json.dump(d, f)

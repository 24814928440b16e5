"""BASELINE configs 3, 4 and 5 at FULL size against the unmodified reference.

tests/golden/full_size.npz holds, for 4096^2 add x20, 16384^2 add x2 and x9, 8192^2 add x3 + drain x5 and x9 on
the synthetic DEMs, what the reference's own runoffs()/runoffd()/drain() produce (tests/golden/make_golden.py,
run where /root/reference exists): sha256 of the padded water raster, an 8-byte hash of every row (says
which rows differ when the whole hash does), sampled rows, max diff, totaldrain and the sequential
volume sum.  Bit-exact is the bar: every comparison below is on hashes of the fp64 bits.

What is pinned, exactly: one context at 2 / 20 iterations (both kernels, DEM codes on and off, another chunk
height); row blocks at 16384^2 with exchange interval k = 1 (4 slabs) and k = 2 (8 slabs) over 2 iterations, and
with k = 4 AND the default k = 8 on 8 slabs over 9 iterations - halo refreshes after iterations 4 and 8, or after 8,
and a ninth iteration that runs on the refreshed halos - once over peer copies and once over the stand-in RCCL
(tests/mock_rccl); drain at 8192^2 on 1 and 3 slabs (k = 2, 5 iterations) and on 8 slabs at k = 4 and the default
k = 8 over 9 iterations, again over both transports."""
import hashlib
import json
import os

import numpy as np
import pytest

import wdpm_amd
from conftest import GOLDEN
from helpers import pad, sha

MISS, THRES = -99999.0, 0.005 / 1000


@pytest.fixture(scope="module")
def golden():
    z = np.load(os.path.join(GOLDEN, "full_size.npz"))
    return z, {m["name"]: m for m in json.loads(bytes(z["index_json"]).decode())}


def row_hashes(w):
    return np.array([np.frombuffer(hashlib.sha256(r.tobytes()).digest()[:8], dtype=np.uint64)[0] for r in w],
                    dtype=np.uint64)


def assert_matches(z, meta, w):
    name = meta["name"]
    if sha(w) != meta["sha256"]:
        bad = np.nonzero(row_hashes(w) != z[name + "_rowhash"])[0]
        raise AssertionError(f"{name}: {bad.size} rows differ from the reference, first {bad[:8].tolist()}")
    assert np.array_equal(w[::meta["sample_every"]].view(np.uint64), z[name + "_rows"].view(np.uint64))


def inputs(lib, n):
    dem = lib.synth_dem(n, n)
    return pad(dem, np.full((n, n), 0.1), MISS)


def run_add(lib, meta, **ctx_kw):
    n = meta["n"]
    bd, bw = inputs(lib, n)
    with lib.context(module="add", nrows=n, ncols=n, missingvalue=MISS, **ctx_kw) as c:
        c.upload(bd, bw)
        return c, c.run_block(meta["add_iters"], THRES), c.download_water()


def test_oracle_is_pinned_at_4096(oracle, golden):
    """the CPU restatement against the reference at config 3's full size (the small golden vectors cover at
    most two 171-column strips; this covers 24)"""
    z, idx = golden
    meta = idx["cfg3_add_4096_i20"]
    _, md, w = run_add(oracle, meta)
    assert md == meta["max_diff"]
    assert_matches(z, meta, w)


@pytest.mark.gpu
@pytest.mark.parametrize("name,variant", [("cfg3_add_4096_i20", "fused"), ("cfg3_add_4096_i20", "fused-fp64dem"),
                                          ("cfg3_add_4096_i20", "pass"), ("cfg3_add_4096_i20", "fused-chunk48"),
                                          ("cfg4_add_16384_i2", "fused"), ("cfg4_add_16384_i2", "fused-fp64dem")])
def test_add_at_full_size_equals_the_reference(hip, golden, name, variant):
    z, idx = golden
    meta = idx[name]
    n = meta["n"]
    bd, bw = inputs(hip, n)
    kw = dict(kernel=wdpm_amd.KERNEL_PASS) if variant == "pass" else dict(kernel=wdpm_amd.KERNEL_FUSED)
    if variant == "fused-chunk48":
        kw["chunk_rows"] = 48
    with hip.context(module="add", nrows=n, ncols=n, missingvalue=MISS, **kw) as c:
        c.upload(bd, bw)
        if variant == "fused-fp64dem":
            c.set_option(wdpm_amd.OPT_DEM32, 0)
        elif variant.startswith("fused"):
            assert c.get_option(wdpm_amd.OPT_DEM32) == 1       # the synthetic DEMs are decimal: the codes are in use
        md = c.run_block(meta["add_iters"], THRES)
        w = c.download_water()
    assert md == meta["max_diff"]
    assert_matches(z, meta, w)


@pytest.mark.gpu
@pytest.mark.parametrize("n,blocks", [(4096, (6, 14, 9)), (8192, (5, 9))])
def test_chunk_heights_by_xcd_engage_and_change_nothing(hip, monkeypatch, n, blocks):
    """round 4: from the second block of a mostly wet raster on, the marching kernel's chunk heights follow what each XCD delivers
    (wdpm_kernels.h::XcdBalance: weights measured on the device, a table of row boundaries per strip).  The balance must ENGAGE on
    rasters of this size (wdpm_balance_info: at least one rebalance, weights around 1) and must not change a bit of the result:
    the same blocks with WDPM_BALANCE=0 (read when a context is made)."""
    bd, bw = inputs(hip, n)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("WDPM_BALANCE", mode)
        with hip.context(module="add", nrows=n, ncols=n, missingvalue=MISS) as c:
            c.upload(bd, bw)
            mds = [c.run_block(k, THRES) for k in blocks]
            updates, weights = c.balance_info()
            out[mode] = (sha(c.download_water()), mds, updates, weights)
    assert out["1"][:2] == out["0"][:2]
    assert out["0"][2] == 0 and out["0"][3] == [1.0] * 8 + [0.95]
    updates, weights = out["1"][2:]
    assert updates >= 1, "the balance never engaged"
    assert all(0.7 <= w <= 1.4 for w in weights[:8]) and abs(sum(weights[:8]) / 8 - 1.0) < 0.02 and 0.75 <= weights[8] <= 1.2, weights
    os.makedirs(os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out", "xcd_weights.txt"), "a") as f:
        f.write(f"{n}x{n} after blocks {blocks}: {updates} rebalances, weights {[round(w, 3) for w in weights]}\n")


@pytest.mark.gpu
@pytest.mark.parametrize("devices,k", [([0, 0, 0, 0], 1), ([0] * 8, 2)])
def test_row_blocks_at_16384_equal_the_reference(hip, golden, devices, k):
    """config 4's decomposition (here: the slabs of one GPU, one host thread each) against the REFERENCE's
    bits, not only against one slab"""
    from wdpm_amd.rowblock import Group
    z, idx = golden
    meta = idx["cfg4_add_16384_i2"]
    n = meta["n"]
    bd, bw = inputs(hip, n)
    with Group(hip, "add", n, n, MISS, devices, exchange_every=k) as g:
        assert g.size == len(devices)
        g.upload(bd, bw)
        md = g.run_block(meta["add_iters"], THRES)
        w = g.download_water()
    assert md == meta["max_diff"]
    assert_matches(z, meta, w)


@pytest.mark.gpu
@pytest.mark.parametrize("every", [None, 4])
def test_row_blocks_at_the_default_exchange_interval_equal_the_reference(hip, golden, every):
    """config 4 as bench.py and WDPMCL run it by default: 8 row blocks, halos refreshed every EIGHT iterations (round 3:
    179 against 187 us per iteration on an 8-GPU slab, profiles/r03/scale_projection.txt; and every four, the earlier
    default); nine iterations, so that the ninth runs on halos refreshed under that schedule (WDPMCL.c:1094-1106)"""
    from wdpm_amd.rowblock import Group
    z, idx = golden
    meta = idx["cfg4_add_16384_i9"]
    n = meta["n"]
    bd, bw = inputs(hip, n)
    with Group(hip, "add", n, n, MISS, [0] * 8, **(dict(exchange_every=every) if every else {})) as g:
        assert g.size == 8
        import ctypes as C
        k = C.c_int32()
        hip.check(hip.dll.wdpm_rank_info(hip.dll.wdpm_group_rank(g._h, 0), None, C.byref(k), None))
        assert k.value == (every or 8)
        g.upload(bd, bw)
        md = g.run_block(meta["add_iters"], THRES)
        w = g.download_water()
    assert md == meta["max_diff"]
    assert_matches(z, meta, w)


@pytest.mark.gpu
def test_default_schedule_over_the_standin_rccl_equals_the_reference(hip):
    """the same two full-size jobs (add 16384^2 x9, drain 8192^2 x9; 8 slabs, default k = 8) with the halos going through the
    library's RCCL path (wdpm_comm_exchange on each rank's stream; the wire is tests/mock_rccl) - in a child process,
    because the stand-in has to be bound before the library looks for RCCL"""
    import subprocess
    import sys
    from conftest import ROOT
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "mock_rccl")], stdout=subprocess.DEVNULL)
    env = dict(os.environ, WDPM_RCCL_LIB=os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so"), WDPM_HALO="rccl",
               WDPM_RCCL_SHARED_DEVICE_OK="1", WDPM_FULL_SIZE_WORKER="1",
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")]))
    p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "FULL_SIZE_RCCL_OK add" in p.stdout and "FULL_SIZE_RCCL_OK drain" in p.stdout, p.stdout


def drain_job(hip, z, meta, devices, k):
    from wdpm_amd.rowblock import Group
    n = meta["n"]
    bd, bw = inputs(hip, n)
    with hip.context(module="add", nrows=n, ncols=n, missingvalue=MISS) as c:
        c.upload(bd, bw)
        c.iterate(meta["add_iters"])
        w3 = c.download_water()
    dr, dc = meta["drainrow"], meta["draincol"]
    kw = dict(exchange_every=k) if k else {}
    with Group(hip, "drain", n, n, MISS, devices, drainrow=dr, draincol=dc, **kw) as g:
        assert g.size == len(devices)
        g.upload(bd, w3)
        g.set_totaldrain(meta["td0"])
        md = g.run_block(meta["drain_iters"], THRES)
        diffdrain, vol = g.drain_stats()
        td = g.totaldrain()
        w = g.download_water()
        halo = g.halo_kind
    assert md == meta["max_diff"] and td == meta["totaldrain"]
    assert diffdrain == abs(meta["totaldrain"] - meta["td0"])
    assert vol == meta["volume_sum"]
    assert_matches(z, meta, w)
    return halo


@pytest.mark.gpu
@pytest.mark.parametrize("every", [None, 4])
def test_drain_on_8_slabs_at_the_default_interval_equals_the_reference(hip, golden, every):
    """config 5's decomposition: 8 row blocks, default exchange interval (8) and 4, nine drain iterations (WDPMCL.c:1076-1093)"""
    z, idx = golden
    drain_job(hip, z, idx["cfg5_drain_8192_a3_d9"], [0] * 8, every)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_drain_at_8192_equals_the_reference(hip, golden, devices):
    """config 5 at full size: add x3, flush, drain x5 - water, max diff, totaldrain and the sequential
    volume sum; 8192^2 is where the drain kernel's two-waves-per-SIMD launch shape switches on"""
    from wdpm_amd.rowblock import Group
    z, idx = golden
    meta = idx["cfg5_drain_8192_a3_d5"]
    n = meta["n"]
    bd, bw = inputs(hip, n)
    with hip.context(module="add", nrows=n, ncols=n, missingvalue=MISS) as c:
        c.upload(bd, bw)
        c.iterate(meta["add_iters"])
        w3 = c.download_water()
    dr, dc = meta["drainrow"], meta["draincol"]
    k = int(np.argmin(np.where(bd > 0, bd, np.inf)))
    assert (k // (n + 2), k % (n + 2)) == (dr, dc)
    assert max(float(w3[dr, dc]), 0.0) == meta["td0"]
    with Group(hip, "drain", n, n, MISS, devices, exchange_every=2, drainrow=dr, draincol=dc) as g:
        g.upload(bd, w3)
        g.set_totaldrain(meta["td0"])
        md = g.run_block(meta["drain_iters"], THRES)
        diffdrain, vol = g.drain_stats()
        td = g.totaldrain()
        w = g.download_water()
    assert md == meta["max_diff"] and td == meta["totaldrain"]
    assert diffdrain == abs(meta["totaldrain"] - meta["td0"])
    assert vol == meta["volume_sum"]
    assert_matches(z, meta, w)


if __name__ == "__main__" and os.environ.get("WDPM_FULL_SIZE_WORKER"):
    # child of test_default_schedule_over_the_standin_rccl_equals_the_reference (WDPM_RCCL_LIB points at the stand-in)
    import wdpm_amd as _w
    from wdpm_amd.rowblock import Group as _Group
    _hip = _w.load_hip()
    assert b"2.99.99" in _hip.dll.wdpm_comm_version()
    _z = np.load(os.path.join(GOLDEN, "full_size.npz"))
    _idx = {m["name"]: m for m in json.loads(bytes(_z["index_json"]).decode())}
    _meta = _idx["cfg4_add_16384_i9"]
    _bd, _bw = inputs(_hip, _meta["n"])
    with _Group(_hip, "add", _meta["n"], _meta["n"], MISS, [0] * 8) as _g:
        assert _g.size == 8 and _w.HALO_NAMES[_g.halo_kind] == "rccl"
        _g.upload(_bd, _bw)
        _md = _g.run_block(_meta["add_iters"], THRES)
        _wat = _g.download_water()
    assert _md == _meta["max_diff"]
    assert_matches(_z, _meta, _wat)
    del _bd, _bw, _wat
    print("FULL_SIZE_RCCL_OK add", flush=True)
    assert _w.HALO_NAMES[drain_job(_hip, _z, _idx["cfg5_drain_8192_a3_d9"], [0] * 8, None)] == "rccl"
    print("FULL_SIZE_RCCL_OK drain", flush=True)

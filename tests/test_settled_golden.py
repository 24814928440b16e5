"""The BASELINE configurations in the regime they are defined in, at FULL size, against the unmodified reference.

tests/test_full_size_golden.py pins 2 - 20 iterations from a uniform 0.1 m sheet.  SURVEY 8d words the configurations otherwise:
config 3 is "exactly 1000 iterations (one block)" at 4096^2, config 4 the same at 16384^2, config 5 drains 8192^2 from "that
raster's add-100-mm state after 1000 iterations".  tests/golden/full_size.npz holds (tests/golden/make_golden.py `settled`, hours
of the reference's own runoffs() / runoffd() / drain() on the build container's cores):

  cfg3_add_4096_i1000, cfg3_add_4096_b2_i2000      one block of 1000, then a SECOND block behind its threshold flush (WDPMCL.c:1055-1065)
  cfg4_add_16384_i100 (_i300, _i1000)             one block of 100 (300, 1000) iterations of the metric's own raster
  cfg5_add_8192_i1000, cfg5_drain_8192_a1000_d100, _d1000    config 5 as worded: the add state, then ONE drain block seen after 100 and 1000
  cfg3x_ponds_4096_i200, cfg3x_ponds_4096_b2_i400  what the all-wet configurations never reach at full size (after 1000 iterations every
                                                  cell is still wet and the deepest pond is 1.1 m): 12 m ponds on a mostly DRY raster -
                                                  flows of 1.5 m, where a clamped neighbour step would be wrong, and dry tiles

Every comparison is on hashes of the fp64 bits (sha256 of the padded raster, an 8-byte hash per row, sampled rows) and on the block
scalars as doubles.  Each job runs as the library dispatches it by default (marching kernel, two waves per SIMD, DEM codes, XCD balance
rebuilding its table during the block) on ONE context, with the fp64 DEM, on EIGHT row blocks at the default exchange interval over
peer copies, and - in a child process - on eight row blocks over the library's RCCL path on the stand-in wire (tests/mock_rccl);
tests/test_multi_gpu.py runs the per-device variants on a multi-GPU lease.  An entry that has not been generated yet is skipped by name."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import wdpm_amd
from conftest import GOLDEN, ROOT
from helpers import sha
from test_full_size_golden import MISS, THRES, assert_matches, inputs

pytestmark = pytest.mark.gpu


def load_golden():
    z = np.load(os.path.join(GOLDEN, "full_size.npz"))
    return z, {m["name"]: m for m in json.loads(bytes(z["index_json"]).decode())}


@pytest.fixture(scope="module")
def golden():
    return load_golden()


def need(idx, *names):
    missing = [n for n in names if n not in idx]
    if missing:
        pytest.skip(f"tests/golden/full_size.npz has no {missing} yet (tests/golden/make_golden.py settled ...)")
    return [idx[n] for n in names]


def ponds(n):
    """water-in of cfg3x (tests/golden/make_golden.py): 12 m on one 256 x 256 block in eight, file coordinates"""
    bi, bj = np.mgrid[0:n, 0:n] // 256
    return np.where((3 * bi + 5 * bj) % 8 == 0, 12.0, 0.0)


def group(lib, module, n, devices, fp64dem=False, **kw):
    from wdpm_amd.rowblock import Group
    g = Group(lib, module, n, n, MISS, devices, **kw)
    assert g.size == len(devices)
    if fp64dem:
        for i in range(g.size):
            lib.check(lib.dll.wdpm_set_option(g.rank_ctx(i), wdpm_amd.OPT_DEM32, 0))
    return g


def tiles(lib, g):
    import ctypes as C
    seen = worked = 0
    for i in range(g.size):
        for key in (wdpm_amd.capi.OPT_TILES_SEEN, wdpm_amd.capi.OPT_TILES_WORKED):
            v = C.c_int64()
            lib.check(lib.dll.wdpm_get_option(g.rank_ctx(i), key, C.byref(v)))
            if key == wdpm_amd.capi.OPT_TILES_SEEN:
                seen += v.value
            else:
                worked += v.value
    return seen, worked


# ------------------------------------------------------------------------------------------------------------ the jobs
def job_two_blocks(lib, z, idx, first, second, devices, water=None, fp64dem=False, want_dry=False):
    """an add job of two blocks from `first`'s start: the state and max diff after each block (the second starts with a threshold flush
    that zeroes `flushed_before_block2` cells: WDPMCL.c:1055-1065)"""
    m1, m2 = idx[first], idx[second]
    n = m1["n"]
    bd, bw = inputs(lib, n)
    if water is not None:
        bw[1:-1, 1:-1] = water(n)
    with group(lib, "add", n, devices, fp64dem) as g:
        g.upload(bd, bw)
        md1 = g.run_block(m1["blocks"][0], THRES)
        w1 = g.download_water()
        assert md1 == m1["max_diff"], (md1, m1["max_diff"])
        assert_matches(z, m1, w1)
        assert int(np.count_nonzero(w1 > 0)) == m1["wet_cells"] and float(w1.max()) == m1["deepest"]
        if m2["flushed_before_block2"]:
            assert int(np.count_nonzero((w1 < THRES) & (w1 != 0))) == m2["flushed_before_block2"]     # the flush has something to do
        del w1
        md2 = g.run_block(m2["blocks"][1], THRES)
        w2 = g.download_water()
        assert md2 == m2["max_diff"], (md2, m2["max_diff"])
        assert_matches(z, m2, w2)
        if want_dry:
            seen, worked = tiles(lib, g)
            return seen, worked
    return None


def job_one_block(lib, z, meta, devices, fp64dem=False):
    n = meta["n"]
    bd, bw = inputs(lib, n)
    with group(lib, "add", n, devices, fp64dem) as g:
        g.upload(bd, bw)
        md = g.run_block(meta["add_iters"], THRES)
        w = g.download_water()
        updates = []
        for i in range(g.size):
            import ctypes as C
            nb, wb = C.c_int32(), (C.c_double * 9)()
            lib.check(lib.dll.wdpm_balance_info(g.rank_ctx(i), C.byref(nb), wb))
            updates.append(nb.value)
    assert md == meta["max_diff"], (md, meta["max_diff"])
    assert_matches(z, meta, w)
    return updates


def job_config5(lib, z, idx, devices, fp64dem=False, upto=(100, 1000)):
    """config 5 as SURVEY 8d words it: add 100 mm x1000 at 8192^2, then ONE drain block from that state - water, max diff, totaldrain,
    |d totaldrain| and the sequential volume sum after 100 and after 1000 iterations (WDPMCL.c:1076-1093, :1257-1268)"""
    ma = idx["cfg5_add_8192_i1000"]
    n = ma["n"]
    bd, bw = inputs(lib, n)
    with group(lib, "add", n, devices, fp64dem) as g:
        g.upload(bd, bw)
        md = g.run_block(1000, THRES)
        w_add = g.download_water()
    assert md == ma["max_diff"]
    assert_matches(z, ma, w_add)
    halo = None
    for k in upto:
        m = idx[f"cfg5_drain_8192_a1000_d{k}"]
        dr, dc = m["drainrow"], m["draincol"]
        assert max(float(w_add[dr, dc]), 0.0) == m["td0"]
        assert int(np.count_nonzero((w_add < THRES) & (w_add != 0))) == m["flushed_before_drain"]
        with group(lib, "drain", n, devices, fp64dem, drainrow=dr, draincol=dc) as g:
            g.upload(bd, w_add)
            g.set_totaldrain(m["td0"])
            md = g.run_block(k, THRES)
            diffdrain, vol = g.drain_stats()
            td = g.totaldrain()
            w = g.download_water()
            halo = g.halo_kind
        assert md == m["max_diff"] and td == m["totaldrain"], (k, md, m["max_diff"], td, m["totaldrain"])
        assert diffdrain == abs(m["totaldrain"] - m["td0"]) and vol == m["volume_sum"], (k, diffdrain, vol, m["volume_sum"])
        assert_matches(z, m, w)
    return halo


# ------------------------------------------------------------------------------------------------------------ one GPU, as dispatched
VARIANTS = [("one", [0], False), ("one-fp64dem", [0], True), ("eight", [0] * 8, False)]


@pytest.mark.parametrize("tag,devices,fp64dem", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_config_3_one_block_of_1000_and_a_second_behind_the_flush(hip, golden, tag, devices, fp64dem):
    z, idx = golden
    need(idx, "cfg3_add_4096_i1000", "cfg3_add_4096_b2_i2000")
    job_two_blocks(hip, z, idx, "cfg3_add_4096_i1000", "cfg3_add_4096_b2_i2000", devices, fp64dem=fp64dem)


CFG4 = [(100, "one", [0], False), (100, "one-fp64dem", [0], True), (100, "eight", [0] * 8, False), (300, "eight", [0] * 8, False),
        (1000, "one", [0], False)]


@pytest.mark.parametrize("iters,tag,devices,fp64dem", CFG4, ids=[f"{v[0]}-{v[1]}" for v in CFG4])
def test_config_4_settling(hip, golden, tag, devices, fp64dem, iters):
    """the metric's own raster, 16384^2, one block of 100 iterations (300 on eight row blocks, the full 1000 of SURVEY's config 4 on
    one context - entries that take the reference hours of a core, skipped by name until they exist)"""
    z, idx = golden
    (meta,) = need(idx, f"cfg4_add_16384_i{iters}")
    updates = job_one_block(hip, z, meta, devices, fp64dem)
    if tag == "one" and os.environ.get("WDPM_BALANCE", "1") not in ("0",):
        assert updates[0] >= 1, "the XCD balance never rebuilt its table during the block: not the regime this test is for"


@pytest.mark.parametrize("tag,devices,fp64dem", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_config_5_drain_from_the_settled_add_state(hip, golden, tag, devices, fp64dem):
    z, idx = golden
    need(idx, "cfg5_add_8192_i1000", "cfg5_drain_8192_a1000_d100")
    job_config5(hip, z, idx, devices, fp64dem, upto=[k for k in (100, 1000) if f"cfg5_drain_8192_a1000_d{k}" in idx])


@pytest.mark.parametrize("tag,devices,fp64dem", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_deep_ponds_on_a_mostly_dry_raster(hip, golden, tag, devices, fp64dem):
    """12 m ponds: a flow of w / 8 = 1.5 m would come out as 1.0 from the clamped neighbour step, so equality with the reference
    here says the depth guard sent those windows down the unclamped path; seven blocks in eight start dry, so the launches that
    keep tile flags must have seen tiles and skipped most of them"""
    z, idx = golden
    m1, _ = need(idx, "cfg3x_ponds_4096_i200", "cfg3x_ponds_4096_b2_i400")
    assert m1["deepest"] > 8.0
    t = job_two_blocks(hip, z, idx, "cfg3x_ponds_4096_i200", "cfg3x_ponds_4096_b2_i400", devices, water=ponds, fp64dem=fp64dem,
                       want_dry=True)
    if os.environ.get("WDPM_TILES", "1") != "0":
        seen, worked = t
        assert seen > 0 and worked < 0.6 * seen, (seen, worked)      # tiles were tracked, and a good part of them skipped


def test_settled_jobs_over_the_standin_rccl(hip, golden):
    """the same jobs on eight row blocks with the halos going through the library's RCCL path (wdpm_comm_exchange on each rank's
    stream, the overlapped last iteration of every group of eight, ncclAllGather for the block scalars) - in a child process,
    because the stand-in has to be bound before the library looks for RCCL"""
    _, idx = golden
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "mock_rccl")], stdout=subprocess.DEVNULL)
    env = dict(os.environ, WDPM_RCCL_LIB=os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so"), WDPM_HALO="rccl",
               WDPM_RCCL_SHARED_DEVICE_OK="1", WDPM_SETTLED_WORKER="1",
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")]))
    p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    for tag, names in (("cfg3", ["cfg3_add_4096_b2_i2000"]), ("cfg4", ["cfg4_add_16384_i100"]), ("cfg5", ["cfg5_drain_8192_a1000_d100"]),
                       ("cfg3x", ["cfg3x_ponds_4096_b2_i400"])):
        if all(n in idx for n in names):
            assert f"SETTLED_RCCL_OK {tag}" in p.stdout, p.stdout


if __name__ == "__main__" and os.environ.get("WDPM_SETTLED_WORKER"):
    _hip = wdpm_amd.load_hip()
    assert b"2.99.99" in _hip.dll.wdpm_comm_version()
    _z, _idx = load_golden()
    _dev = [0] * 8
    if "cfg3_add_4096_b2_i2000" in _idx:
        job_two_blocks(_hip, _z, _idx, "cfg3_add_4096_i1000", "cfg3_add_4096_b2_i2000", _dev)
        print("SETTLED_RCCL_OK cfg3", flush=True)
    if "cfg4_add_16384_i100" in _idx:
        job_one_block(_hip, _z, _idx["cfg4_add_16384_i100"], _dev)
        print("SETTLED_RCCL_OK cfg4", flush=True)
    if "cfg5_drain_8192_a1000_d100" in _idx:
        assert wdpm_amd.HALO_NAMES[job_config5(_hip, _z, _idx, _dev, upto=[k for k in (100, 1000) if f"cfg5_drain_8192_a1000_d{k}" in _idx])] == "rccl"
        print("SETTLED_RCCL_OK cfg5", flush=True)
    if "cfg3x_ponds_4096_b2_i400" in _idx:
        job_two_blocks(_hip, _z, _idx, "cfg3x_ponds_4096_i200", "cfg3x_ponds_4096_b2_i400", _dev, water=ponds)
        print("SETTLED_RCCL_OK cfg3x", flush=True)

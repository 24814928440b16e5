"""Dry-tile skipping (wdpm_kernels.h::TileFlags): rasters that are mostly dry land with water that spreads from a
few places, against the oracle.  The flags must never change a bit of the result - and must actually skip work."""
import numpy as np
import pytest

import wdpm_amd
from helpers import bits_equal, find_drain, n_bit_diff, pad

pytestmark = pytest.mark.gpu
MISS = -99999.0


def dry_case(R, C, seed, nodata=True):
    """a tilted, bumpy DEM; water only in a few blobs high up, so that it runs down across tile boundaries"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:R, 0:C]
    dem = np.round(520.0 - 0.02 * x - 0.03 * y + 0.3 * np.sin(x / 7.0) * np.cos(y / 5.0) + 0.05 * rng.random((R, C)), 4)
    if nodata:
        dem[rng.random((R, C)) < 0.03] = MISS
        dem[R // 2:R // 2 + 20, C // 3:C // 3 + 60] = MISS
    water = np.zeros((R, C))
    for (r, c) in [(10, 20), (R // 3, C // 2), (R // 2 + 30, C - 40)]:
        water[r:r + 9, c:c + 13] = 0.4 * rng.random((9, 13)) + 0.05
    water[dem <= MISS] = 0.0
    return dem, water


def run_pair(hip, oracle, module, R, C, script, chunk, seed=1, tiles=1, sparse=None, **kw):
    dem, water = dry_case(R, C, seed)
    bd, bw = pad(dem, water, MISS)
    ckw = dict(module=module, nrows=R, ncols=C, missingvalue=MISS, **kw)
    if module == "drain":
        dr, dc = find_drain(bd)
        ckw.update(drainrow=dr, draincol=dc)
    with hip.context(kernel=wdpm_amd.KERNEL_FUSED, chunk_rows=chunk, **ckw) as g, oracle.context(**ckw) as o:
        g.set_option(wdpm_amd.capi.OPT_TILES, tiles)
        if sparse is not None:
            g.set_option(wdpm_amd.capi.OPT_SPARSE, sparse)
        for c in (g, o):
            c.upload(bd, bw)
            c.totaldrain = 0.0
        for op in script:
            for c in (g, o):
                if op[0] == "it":
                    c.iterate(op[1])
                elif op[0] == "begin":
                    c.begin_block(op[1])
                elif op[0] == "rows":
                    c.upload_rows(op[1], op[2])
            if op[0] == "check":
                assert n_bit_diff(g.download_water(), o.download_water()) == 0, op
                assert g.max_diff() == o.max_diff()
                if module == "drain":
                    assert g.totaldrain == o.totaldrain
        seen, worked = g.get_option(wdpm_amd.capi.OPT_TILES_SEEN), g.get_option(wdpm_amd.capi.OPT_TILES_WORKED)
        return seen, worked, g.get_option(wdpm_amd.capi.OPT_SPARSE)


@pytest.mark.parametrize("module", ["add", "drain"])
@pytest.mark.parametrize("chunk", [6, 12, 30])
def test_dry_tiles_are_skipped_and_nothing_changes(hip, oracle, module, chunk):
    R, C = 210, 700
    rows = np.zeros((3, C + 2))
    rows[:, 300:340] = 0.2                                   # water arriving from outside into a dry region
    script = [("begin", 1e-4), ("it", 7), ("check",), ("it", 20), ("check",), ("begin", 2e-3), ("it", 15), ("check",),
              ("rows", 150, rows), ("it", 9), ("check",), ("it", 30), ("check",)]
    seen, worked, _ = run_pair(hip, oracle, module, R, C, script, chunk)
    assert seen > 0 and worked < seen * (0.8 if chunk < 30 else 0.95), (seen, worked)   # tiles that never did any work
    seen0, worked0, _ = run_pair(hip, oracle, module, R, C, script, chunk, tiles=0)
    assert seen0 == 0 and worked0 == 0                          # switched off: the flags are not even kept


def test_sparse_mode_marches_short_chunks(hip, oracle):
    """a raster tall enough for the short-chunk mode: forced on, the library's own switch observed, same bits"""
    R, C = 1700, 1000
    script = [("begin", 1e-4), ("it", 12), ("check",), ("begin", 1e-4), ("it", 12), ("check",)]
    seen, worked, sparse = run_pair(hip, oracle, "add", R, C, script, 0, sparse=1)
    assert sparse == 1 and seen > 0 and worked < 0.6 * seen, (seen, worked, sparse)
    # left to itself the library goes sparse at the first look (the end of the first block): most tiles are dry
    seen, worked, sparse = run_pair(hip, oracle, "add", R, C, script, 0)
    assert seen > 0 and sparse == 1, (seen, worked, sparse)


@pytest.mark.parametrize("k", [1, 3])
def test_dry_tiles_across_slabs(hip, oracle, k):
    """several slabs of one GPU with peer-copied halos: rows arriving from a neighbour make tiles wet again"""
    from wdpm_amd.rowblock import Group
    R, C = 420, 520
    dem, water = dry_case(R, C, 4)
    bd, bw = pad(dem, water, MISS)
    res = {}
    for name, lib, devices in (("hip", hip, [0, 0, 0]), ("oracle", oracle, [0])):
        with Group(lib, "add", R, C, MISS, devices, exchange_every=k, chunk_rows=12 if lib is hip else 0) as g:
            g.upload(bd, bw)
            mds = [g.run_block(17, 1e-4), g.run_block(23, 1e-4)]
            res[name] = (g.download_water(), mds)
    assert bits_equal(res["hip"][0], res["oracle"][0]) and res["hip"][1] == res["oracle"][1]


@pytest.mark.parametrize("module", ["add", "drain"])
def test_random_ponds_and_call_sequences(hip, oracle, module):
    """random rasters with a few ponds (so that whole tiles are dry and stay dry, or get wet later), random chunk heights,
    random sequences of begin / iterate / expect / max_diff / upload_rows (water dropped into dry land) / volume / download
    with tile skipping on: every observation equals the oracle's, and over the whole hunt tiles really were skipped"""
    import random
    import os
    lo, hi = (int(v) for v in os.environ.get("WDPM_FUZZ_SEEDS", "0:40").split(":"))
    seen = worked = 0
    for seed in range(lo, hi):
        rng = random.Random(seed * 2 + (module == "drain"))
        nrng = np.random.default_rng(seed)
        R, C = rng.randint(40, 330), rng.randint(150, 1300)
        y, x = np.mgrid[0:R, 0:C]
        dem = np.round(520.0 - 0.02 * x - 0.03 * y + 0.3 * np.sin(x / 7.0) * np.cos(y / 5.0) + 0.05 * nrng.random((R, C)), 4)
        if rng.random() < 0.5:
            dem[nrng.random((R, C)) < 0.03] = MISS
        water = np.zeros((R, C))
        for _ in range(rng.randint(0, 4)):
            r, c = rng.randrange(R), rng.randrange(C)
            water[r:r + rng.randint(1, 12), c:c + rng.randint(1, 40)] = 0.3 * rng.random() + 0.01
        water[dem <= MISS] = 0.0
        bd, bw = pad(dem, water, MISS)
        ckw = dict(module=module, nrows=R, ncols=C, missingvalue=MISS)
        if module == "drain":
            dr, dc = find_drain(bd)
            ckw.update(drainrow=dr, draincol=dc)
        chunk = rng.choice([6, 6, 12, 30, 96, 0])
        with hip.context(kernel=wdpm_amd.KERNEL_FUSED, chunk_rows=chunk, **ckw) as g, oracle.context(**ckw) as o:
            if rng.random() < 0.3:
                g.set_option(wdpm_amd.capi.OPT_SPARSE, 1)
            for c in (g, o):
                c.upload(bd, bw)
                c.totaldrain = 0.0
            for step in range(rng.randint(5, 12)):
                op = rng.choice(["it", "it", "it", "begin", "rows", "maxdiff", "expect", "volume", "download"])
                if op == "it":
                    n = rng.randint(1, 6)
                    g.iterate(n); o.iterate(n)
                elif op == "begin":
                    t = rng.choice([0.0, 1e-3])
                    g.begin_block(t); o.begin_block(t)
                elif op == "rows":
                    n = rng.randint(1, 4); r0 = rng.randint(0, R + 2 - n)
                    rows = np.zeros((n, C + 2))
                    c0 = rng.randrange(C)
                    rows[:, c0:c0 + rng.randint(1, 30)] = 0.2
                    rows = np.where(bd[r0:r0 + n] > MISS, rows, 0.0)
                    g.upload_rows(r0, rows); o.upload_rows(r0, rows)
                elif op == "expect":
                    g.expect_max_diff(0, R + 2); o.expect_max_diff(0, R + 2)
                elif op == "maxdiff":
                    assert g.max_diff() == o.max_diff(), (seed, step)
                elif op == "volume":
                    assert g.volume_partial(0, R + 2, 0.0) == o.volume_partial(0, R + 2, 0.0), (seed, step)
                else:
                    assert n_bit_diff(g.download_water(), o.download_water()) == 0, (seed, step, R, C, chunk)
            assert n_bit_diff(g.download_water(), o.download_water()) == 0, (seed, R, C, chunk)
            assert g.max_diff() == o.max_diff() and g.totaldrain == o.totaldrain
            seen += g.get_option(wdpm_amd.capi.OPT_TILES_SEEN)
            worked += g.get_option(wdpm_amd.capi.OPT_TILES_WORKED)
    assert 0 < worked < 0.8 * seen, (worked, seen)

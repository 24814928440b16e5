"""Set-up and final statistics next to the rasters (wdpm_group_upload_unpadded, count_stats,
find_drain, set_drain, get_cell, download_unpadded) against a numpy restatement of the reference's host loops
(WDPMCL.c:643-650, :727-740, :796-807, :879-885, :1005-1017, :1379-1459), on one slab and on several."""
import numpy as np
import pytest

from helpers import bits_equal, find_drain, pad, random_case
from wdpm_amd.rowblock import Group

CASES = [("add", 1, dict(add=0.07, rof=0.6)), ("subtract", 2, dict(sub=0.05)), ("drain", 0, {})]


def host_setup(dem, water, miss, op, add=0.0, rof=0.0, sub=0.0):
    w = water.copy()
    valid = dem > miss
    if op == 1:                                   # WDPMCL.c:727-740
        wet = valid & (w > 0)
        w[wet] += add
        w[valid & (w <= 0)] = add * rof
    elif op == 2:                                 # :879-885
        w[valid] = np.maximum(w[valid] - sub, 0.0)
    return pad(dem, w, miss)


def check(lib, devices, module, op, kw, with_water, R=97, C=230, seed=303, missing_frac=0.2, exchange_every=2):
    dem, water, miss = random_case(seed, R, C, missing_frac=missing_frac, dry_frac=0.4)
    water[dem <= miss] = miss                     # as in an output raster of a previous run: NODATA cells hold the NODATA value
    if not with_water:
        water = None
    bd, bw = host_setup(dem, np.zeros_like(dem) if water is None else water, miss, op, **kw)
    dr, dc = find_drain(bd)
    gkw = dict(drainrow=-1, draincol=-1) if module == "drain" else {}
    with Group(lib, module, R, C, miss, devices, exchange_every=exchange_every, **gkw) as g:
        g.upload_unpadded(dem, water, op=op, **kw)
        assert bits_equal(g.download_water(), bw)                                 # padded rasters as the host would build them
        valid, wet, mx = g.count_stats()
        v = bd > miss
        assert valid == int(v.sum()) and wet == int((v & (bw > 0.001)).sum())
        assert mx == float(np.where(v, bw, miss).max())
        if (bd > 0).any():
            assert g.find_drain() == (float(bd[dr, dc]), dr, dc)
        else:                                         # no cell above 0: the reference leaves drainrow = draincol = 0 (:1006-1017)
            assert g.find_drain()[1:] == (0, 0) and (dr, dc) == (0, 0)
        assert g.get_cell(dr, dc) == (float(bw[dr, dc]), float(bd[dr, dc]))
        if module == "drain":
            rc = g.set_drain(dr, dc)
            assert rc in (0, 2)
            if rc == 0:
                g.set_totaldrain(max(float(bw[dr, dc]), 0.0))
                g.run_block(3, 1e-5)
        for mask in (True, False):
            w = g.download_water()[1:-1, 1:-1]
            want = np.where(dem > miss, w, miss) if mask else w
            assert bits_equal(g.download_unpadded(mask), want)
        _, vol = g.drain_stats()
        vals = g.download_water()[bd > miss]
        assert vol == (float(np.add.accumulate(vals)[-1]) if len(vals) else 0.0)


@pytest.mark.parametrize("module,op,kw", CASES)
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
@pytest.mark.parametrize("with_water", [True, False])
def test_setup_and_statistics_oracle(oracle, devices, module, op, kw, with_water):
    check(oracle, devices, module, op, kw, with_water)


@pytest.mark.gpu
@pytest.mark.parametrize("module,op,kw", CASES)
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
@pytest.mark.parametrize("with_water", [True, False])
def test_setup_and_statistics_hip(hip, devices, module, op, kw, with_water):
    check(hip, devices, module, op, kw, with_water)


def test_outlet_next_to_a_boundary_asks_for_a_new_partition(oracle):
    """a drain group made before the outlet is known refuses an outlet within three rows of a slab boundary"""
    from wdpm_amd.rowblock import partition
    R, C, miss = 120, 40, -99999.0
    b = partition(oracle, R, 2, 1)[1].own_lo
    dem = np.full((R, C), 500.0)
    with Group(oracle, "drain", R, C, miss, [0, 0], exchange_every=1, drainrow=-1, draincol=-1) as g:
        g.upload_unpadded(dem, None)
        assert g.set_drain(b + 1, 7) == 2 and g.set_drain(b - 2, 7) == 2
        assert g.set_drain(b + 3, 7) == 0 and g.set_drain(b - 4, 7) == 0


def random_shapes(lib, seeds):
    """the same checks on random shapes (1 x 1 ... 150 x 300), 1 - 6 slabs, any module, with and without a water file"""
    import random
    for seed in seeds:
        rng = random.Random(seed)
        module, op, kw = CASES[rng.randrange(3)]
        R, C = rng.choice([(1, 1), (2, 5), (rng.randint(3, 150), rng.randint(1, 300))])
        check(lib, [0] * rng.randint(1, 6), module, op, kw, rng.random() < 0.5, R=R, C=C, seed=seed,
              missing_frac=rng.choice([0.0, 0.2, 0.7]), exchange_every=rng.randint(1, 5))


def test_setup_and_statistics_random_shapes_oracle(oracle):
    random_shapes(oracle, range(100))


@pytest.mark.gpu
def test_setup_and_statistics_random_shapes_hip(hip):
    import os
    lo, hi = (int(v) for v in os.environ.get("WDPM_FUZZ_SEEDS", "100:180").split(":"))
    random_shapes(hip, range(lo, hi))

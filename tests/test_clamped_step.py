"""The clamped neighbour step (round 4; wdpm_stencil.h::eighth_clamped, `deep` in wdpm_fused.hip).

`max(x / 8, -0.0)` of the neighbour step (WDPMCL.c:1947-1957) is ONE instruction on gfx950 - `v_ldexp_f64 f, x, -3 clamp` -
as long as the flow stays below 1 m; the kernels take it only where the depths they hold guarantee that: every depth of a
wave's window below 3.000002 m when it was loaded (and every valid elevation below 2^30 m in magnitude), because

    a flow is at most (the centre's depth + half an ulp of its elevation) / 8, a cell receives in at most eight of an
    iteration's nine passes, in one block per pass: depths grow by at most (9/8)^8 = 2.566 within an iteration.

CPU part: the growth bound, checked on the oracle's own passes (the reference's arithmetic) with rasters built to make depths
pile up; the numpy model of the instruction against the two-instruction form.  GPU part (-m gpu): rasters whose depths sit
on both sides of the 3 m and 8 m lines, through every kernel family, bit for bit against the oracle."""
import numpy as np
import pytest

import wdpm_amd
from helpers import find_drain, n_bit_diff, pad
from test_stencil_forms import nz_drain_step, nz_step, vmax, vmin

GROWTH = (9.0 / 8.0) ** 8


def eighth_clamped(x):
    """v_ldexp_f64 x, -3 clamp as tools/clamp_probe.hip measured it on the chip: clamp(x / 8) to [0, 1] after rounding, NaN -> +0.0"""
    with np.errstate(invalid="ignore", over="ignore", under="ignore"):
        q = x * 0.125
        r = np.where(q > 1.0, 1.0, np.where(q > 0.0, q, 0.0))
    return np.where(np.isnan(x), 0.0, r)


def clamped_step(dc, wc, dn, wn, gate, nvalid):
    with np.errstate(invalid="ignore", over="ignore"):
        dce = np.where(gate, dc, -np.inf)
        dnn = np.where(nvalid, dn, np.inf)
        en = dnn + wn
        ht = (dce + wc) - en
        x = np.where(dce > en, wc, ht)
        f = eighth_clamped(x)
        return wc - np.abs(f), wn + f


def clamped_drain_step(dc, wc, dn, wn, gate, nvalid):
    with np.errstate(invalid="ignore", over="ignore"):
        dce = np.where(gate, dc, -np.inf)
        dnn = np.where(nvalid, dn, np.inf)
        wcl = np.where(gate, wc, 0.0)
        nwe = dnn + wn
        ht = (dce + wcl) - nwe
        s = (dce - dnn) + (wcl - wn)
        big = np.where(ht > 0, np.inf, np.where(ht < 0, -np.inf, ht))
        x = np.where(dce > nwe, wcl, vmin(s, big))
        f = eighth_clamped(x)
        return np.where(gate, wcl - np.abs(f), wc), wn + f


def _pools(rng, n):
    dpool = np.array([1e-4, 0.5, 1.0, 499.9999, 500.0, np.nextafter(500.0, 501), 500.0001, 500.1, 502.9, 507.99, 512.0, 1e6, 2.0 ** 30 - 1])
    wpool = np.array([0.0, 0.0, 5e-324, 1e-310, 2.3e-308, 1e-300, 1e-16, 1.1368683772161603e-13, 1e-4, 0.1, np.nextafter(0.1, 1),
                      0.8, 1.0, 2.999, 3.0, 6.0, 7.69, 7.7])      # what the guard admits: < 7.7 m
    dc = dpool[rng.integers(0, len(dpool), n)]
    dn = np.where(rng.random(n) < 0.4, dc, dpool[rng.integers(0, len(dpool), n)])
    wc = wpool[rng.integers(0, len(wpool), n)]
    wn = np.where(rng.random(n) < 0.3, wc, wpool[rng.integers(0, len(wpool), n)])
    jitter = rng.random(n) < 0.3
    wc = np.where(jitter, wc * (1 + rng.normal(0, 1e-15, n)), wc)
    wc = np.minimum(np.abs(wc), 7.7)
    return dc, wc, dn, wn, rng.random(n) < 0.9, rng.random(n) < 0.9


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_clamped_form_equals_the_two_instruction_form_below_8_m(seed):
    rng = np.random.default_rng(seed)
    dc, wc, dn, wn, gate, nvalid = _pools(rng, 2_000_000)
    for plain, clamped in ((nz_step, clamped_step), (nz_drain_step, clamped_drain_step)):
        a = plain(dc, wc, dn, wn, gate & (wc > 0), nvalid)
        b = clamped(dc, wc, dn, wn, gate & (wc > 0), nvalid)
        for u, v in zip(a, b):
            assert np.array_equal(np.asarray(u).view(np.uint64), np.asarray(v).view(np.uint64))


def test_depths_grow_by_less_than_the_bound_within_an_iteration(oracle):
    """The reference's arithmetic (the oracle's single colour passes), rasters built to pile water up: pits under plateaus that are
    all 3 m deep in water, flat quantised ground (where fl(dem + w) - dem rounds a flow up), elevations up to 2^30 m.  After every
    pass and after the iteration: no depth above 3 (9/8)^k, and never 8 m."""
    rng = np.random.default_rng(7)
    R, C, miss, M = 60, 90, -99999.0, 3.0
    for case in range(6):
        base = [500.0, 500.0, 2.0 ** 30 - 4096.0, 0.5, 500.0, 123456.7891][case]
        dem = base + np.round(rng.random((R, C)) * [0.0, 40.0, 2000.0, 0.4, 3.0, 10.0][case], 4)
        pits = rng.random((R, C)) < 0.12
        dem = np.where(pits, dem - 50.0 if base > 100 else dem * 0.01, dem)
        dem[rng.random((R, C)) < 0.03] = miss
        water = np.where(dem > miss, M, 0.0)
        if case == 4:
            water = np.where(rng.random((R, C)) < 0.5, water, water * rng.random((R, C)))
        bd, bw = pad(dem, water, miss)
        with oracle.context(module="add", nrows=R, ncols=C, missingvalue=miss) as o:
            o.upload(bd, bw)
            k = 0
            for oi in (1, 2, 3):
                for oj in (1, 2, 3):
                    o.single_pass(oi, oj)
                    k += 1
                    w = o.download_water()
                    assert w.max() <= M * (9.0 / 8.0) ** min(k, 8) * (1 + 1e-9) + 1e-3, (case, oi, oj, w.max())
            assert o.download_water().max() < M * GROWTH + 1e-3 < 7.7


# ---------------------------------------------------------------------------------------------------------------- GPU


def _deep_case(seed, R, C, module):
    """mostly shallow water with ponds on both sides of the two lines that matter: 3 m (the guard) and 8 m (where the clamp bites)"""
    rng = np.random.default_rng(seed)
    miss = -99999.0
    y, x = np.mgrid[0:R, 0:C]
    dem = 500.0 + 6.0 * np.sin(x / 9.1) * np.cos(y / 7.3) + rng.normal(0, 0.3, (R, C)) - 0.01 * (x + y)
    dem = np.round(dem, 4)
    water = np.where(rng.random((R, C)) < 0.3, 0.0, 0.4 * rng.random((R, C)))
    depths = [2.9, 2.9999999, 3.0, 3.0000019073486324, 3.0000019073486333, 3.1, 5.0, 7.6, 7.9999, 8.0, np.nextafter(8.0, 9), 8.5, 20.0, 100.0]
    for d in depths * 3:
        r, c = int(rng.integers(0, R)), int(rng.integers(0, C))
        h, w = int(rng.integers(1, 12)), int(rng.integers(1, 40))
        water[r:r + h, c:c + w] = d
    # a deep pit under a plateau of 2.999 m: everything around pours in at the guard's limit
    r, c = R // 2, C // 3
    dem[r - 2:r + 3, c - 2:c + 3] = 560.0
    dem[r, c] = 470.0
    water[r - 2:r + 3, c - 2:c + 3] = 2.999
    dem[rng.random((R, C)) < 0.03] = miss
    water = np.where(dem > miss, water, 0.0)
    return dem, water, miss


@pytest.mark.gpu
@pytest.mark.parametrize("module", ["add", "drain"])
@pytest.mark.parametrize("R,C,chunk", [(200, 700, 0), (200, 700, 12), (301, 400, 48), (64, 1100, 30), (900, 500, 0)])
def test_depths_on_both_sides_of_the_guard(hip, oracle, module, R, C, chunk):
    dem, water, miss = _deep_case(R * 7 + C + chunk, R, C, module)
    bd, bw = pad(dem, water, miss)
    kw = dict(module=module, nrows=R, ncols=C, missingvalue=miss)
    if module == "drain":
        dr, dc = find_drain(bd)
        kw.update(drainrow=dr, draincol=dc)
    with hip.context(kernel=wdpm_amd.KERNEL_FUSED, chunk_rows=chunk, **kw) as g, oracle.context(**kw) as o:
        for c in (g, o):
            c.upload(bd, bw)
        if module == "add":
            g.set_option(wdpm_amd.OPT_DEM32, 2 if (R + C) % 2 else 0)
        for n in (1, 2, 7, 40):
            g.iterate(n)
            o.iterate(n)
            assert n_bit_diff(g.download_water(), o.download_water()) == 0, (module, R, C, chunk, n)
            assert g.totaldrain == o.totaldrain
        assert g.run_block(25, 1e-5) == o.run_block(25, 1e-5)
        assert n_bit_diff(g.download_water(), o.download_water()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("module", ["add", "drain"])
def test_huge_elevations_switch_the_clamp_off(hip, oracle, module):
    """half an ulp of a 2^40 m elevation is a tenth of a millimetre: the host keeps the clamped step away from such a DEM"""
    R, C = 150, 420
    dem, water, miss = _deep_case(5, R, C, module)
    dem = np.where(dem > miss, dem + 2.0 ** 40, dem)
    water = np.where(dem > miss, np.minimum(water, 2.9), 0.0)
    bd, bw = pad(dem, water, miss)
    kw = dict(module=module, nrows=R, ncols=C, missingvalue=miss)
    if module == "drain":
        dr, dc = find_drain(bd)
        kw.update(drainrow=dr, draincol=dc)
    for chunk in (0, 12):
        with hip.context(kernel=wdpm_amd.KERNEL_FUSED, chunk_rows=chunk, **kw) as g, oracle.context(**kw) as o:
            for c in (g, o):
                c.upload(bd, bw)
            for n in (1, 9):
                g.iterate(n)
                o.iterate(n)
                assert n_bit_diff(g.download_water(), o.download_water()) == 0
                assert g.totaldrain == o.totaldrain

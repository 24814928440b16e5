"""The 10-instruction neighbour step (wdpm_stencil.h::flow_add_nz: no `ht>0` compare, no min(),
gate carried by the centre elevation) equals the reference's conditional form (WDPMCL.c:1945-1959)
bit for bit on adversarial operands, whenever the neighbour depth is not -0.0.  numpy model of both
forms, including AMD's v_max_f64 treatment of signed zeros and NaN."""
import numpy as np


def vmax(a, b):
    """v_max_f64 (IEEE maxNum, -0 < +0)"""
    with np.errstate(invalid="ignore"):
        r = np.where(a >= b, a, b)
    both_zero = (a == 0) & (b == 0)
    r = np.where(both_zero, np.where(np.signbit(a) & np.signbit(b), -0.0, 0.0), r)
    r = np.where(np.isnan(a), b, np.where(np.isnan(b), a, r))
    return r


def vmin(a, b):
    """v_min_f64 (IEEE minNum, -0 < +0)"""
    with np.errstate(invalid="ignore"):
        r = np.where(a <= b, a, b)
    both_zero = (a == 0) & (b == 0)
    r = np.where(both_zero, np.where(np.signbit(a) | np.signbit(b), -0.0, 0.0), r)
    r = np.where(np.isnan(a), b, np.where(np.isnan(b), a, r))
    return r


def reference_drain_step(dc, wc, dn, wn, gate, nvalid):
    """runoffd()'s non-outlet branch for one neighbour (WDPMCL.c:1977-2000), macros as a<b?a:b / a>b?a:b"""
    with np.errstate(invalid="ignore", over="ignore"):
        cwe = dc + wc
        nwe = dn + wn
        ht = cwe - nwe
        go = gate & nvalid & (ht > 0)
        flow = np.where(dc > nwe, wc / 8.0, ((dc - dn) + (wc - wn)) / 8.0)
        flow = np.where(flow > 0.0, flow, 0.0)
        flow = np.where(flow < wc, flow, wc)
        wc2 = wc - flow
        wc2 = np.where(wc2 > 0.0, wc2, 0.0)
        return np.where(go, wc2, wc), np.where(go, wn + flow, wn)


def nz_drain_step(dc, wc, dn, wn, gate, nvalid):
    """wdpm_stencil.h::flow_drain_nz"""
    with np.errstate(invalid="ignore", over="ignore"):
        dce = np.where(gate, dc, -np.inf)
        dnn = np.where(nvalid, dn, np.inf)
        wcl = np.where(gate, wc, 0.0)           # the block runs on a local centre depth (wdpm_fused.hip)
        nwe = dnn + wn
        ht = (dce + wcl) - nwe
        s = (dce - dnn) + (wcl - wn)
        big = np.where(ht > 0, np.inf, np.where(ht < 0, -np.inf, ht))      # v_ldexp_f64(ht, 2200)
        x = np.where(dce > nwe, wcl, vmin(s, big))
        f = vmax(x * 0.125, np.full_like(x, -0.0))          # (the reference's min(flow, w_c) is dead: wdpm_stencil.h, tests/test_step_floor.py)
        return np.where(gate, wcl - np.abs(f), wc), wn + f


def reference_step(dc, wc, dn, wn, gate, nvalid):
    """runoffs() for one neighbour; gate = centre test (:1099), nvalid = bigdem[n] > missing (:1944)"""
    with np.errstate(invalid="ignore", over="ignore"):
        en = dn + wn
        ht = (dc + wc) - en
        go = gate & nvalid & (ht > 0)
        flow = np.where(dc > en, wc / 8.0, ht / 8.0)
        flow = np.where(flow < wc, flow, wc)
        return np.where(go, wc - flow, wc), np.where(go, wn + flow, wn)


def nz_step(dc, wc, dn, wn, gate, nvalid):
    with np.errstate(invalid="ignore", over="ignore"):
        dce = np.where(gate, dc, -np.inf)
        dnn = np.where(nvalid, dn, np.inf)
        en = dnn + wn
        ht = (dce + wc) - en
        x = np.where(dce > en, wc, ht)
        f = vmax(x * 0.125, np.full_like(x, -0.0))
        return wc - np.abs(f), wn + f


def operands(rng, n):
    ulp = lambda v: np.spacing(np.abs(v))
    dc = rng.choice([1.0, -1.0], n) * 10.0 ** rng.uniform(-3, 4, n)
    dc = np.where(rng.random(n) < 0.1, np.round(dc, 4), dc)
    kind = rng.integers(0, 8, n)
    wc = np.select([kind == 0, kind == 1, kind == 2, kind == 3, kind == 4],
                   [np.zeros(n), rng.integers(1, 50, n) * 5e-324, 10.0 ** rng.uniform(-320, -290, n),
                    rng.integers(0, 9, n) * ulp(dc) / 4, rng.integers(0, 40, n) * ulp(dc)],
                   10.0 ** rng.uniform(-18, 3, n))
    wc = np.where(rng.random(n) < 0.03, -wc, wc)           # negative / -0.0 centres must be no-ops
    k2 = rng.integers(0, 6, n)
    dn = np.select([k2 == 0, k2 == 1, k2 == 2, k2 == 3],
                   [dc, dc + rng.integers(-6, 7, n) * ulp(dc), dc + wc, dc - 10.0 ** rng.uniform(-12, 1, n)],
                   dc + rng.normal(0, 1, n) * 10.0 ** rng.uniform(-14, 2, n))
    k3 = rng.integers(0, 6, n)
    wn = np.select([k3 == 0, k3 == 1, k3 == 2, k3 == 3],
                   [np.zeros(n), wc, rng.integers(0, 9, n) * ulp(dn) / 4, (dc + wc) - dn],
                   10.0 ** rng.uniform(-18, 3, n))
    wn = np.where(rng.random(n) < 0.02, -np.abs(wn) - 1e-300, wn)   # negative depths from odd input files
    wn = np.where((wn == 0) & np.signbit(wn), 0.0, wn)               # the documented precondition: no -0.0
    wn = np.where(rng.random(n) < 0.005, np.nan, wn)                 # NaN cells of odd input files
    wc = np.where(rng.random(n) < 0.005, np.nan, wc)
    gate = (wc > 0.0) & (rng.random(n) < 0.95)
    nvalid = rng.random(n) < 0.93
    return dc, wc, dn, wn, gate, nvalid


def test_nz_form_equals_reference_form():
    rng = np.random.default_rng(7)
    total = 0
    for _ in range(12):
        dc, wc, dn, wn, gate, nvalid = operands(rng, 2_000_000)
        a = reference_step(dc, wc, dn, wn, gate, nvalid)
        b = nz_step(dc, wc, dn, wn, gate, nvalid)
        for u, v, name in ((a[0], b[0], "centre"), (a[1], b[1], "neighbour")):
            bad = u.view(np.uint64) != v.view(np.uint64)
            assert not bad.any(), (name, int(bad.sum()), dc[bad][:3], wc[bad][:3], dn[bad][:3], wn[bad][:3],
                                   u[bad][:3], v[bad][:3])
        total += len(dc)
    assert total >= 2e7


def test_min_is_a_noop_when_water_moves():
    """claim (b) of flow_add_nz on its own: ht > 0 and w_c > 0 imply flow <= w_c"""
    rng = np.random.default_rng(8)
    for _ in range(6):
        dc, wc, dn, wn, gate, nvalid = operands(rng, 2_000_000)
        with np.errstate(invalid="ignore", over="ignore"):
            en = dn + wn
            ht = (dc + wc) - en
            flow = np.where(dc > en, wc / 8.0, ht / 8.0)
            go = (wc > 0) & (ht > 0)
        assert (flow[go] <= wc[go]).all()


def test_nz_drain_form_equals_reference_form():
    """flow_drain_nz (15 instructions: the sign of ht carried by ldexp, no select on the updates,
    the outer max and the min(flow, w_c) dropped) against runoffd()'s conditional form.  The drain sweep only calls it for
    centres with w_c > 0 (WDPMCL.c:1081); negative neighbour depths from odd input files included."""
    rng = np.random.default_rng(17)
    total = 0
    for _ in range(12):
        dc, wc, dn, wn, gate, nvalid = operands(rng, 2_000_000)
        a = reference_drain_step(dc, wc, dn, wn, gate, nvalid)
        b = nz_drain_step(dc, wc, dn, wn, gate, nvalid)
        for u, v, name in ((a[0], b[0], "centre"), (a[1], b[1], "neighbour")):
            bad = u.view(np.uint64) != v.view(np.uint64)
            assert not bad.any(), (name, int(bad.sum()), dc[bad][:3], wc[bad][:3], dn[bad][:3], wn[bad][:3],
                                   u[bad][:3], v[bad][:3])
        total += len(dc)
    assert total >= 2e7

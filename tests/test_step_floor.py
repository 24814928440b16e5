"""Is the neighbour step at its floor?  The search, written down.

The drain step (runoffd's non-outlet branch, WDPMCL.c:1988-2000) costs 14 VALU instructions on gfx950 since round 4
(wdpm_stencil.h::flow_drain_nz<CLAMP>; 16 in rounds 2 - 3, 21 as the reference writes it), the add step (WDPMCL.c:1945-1959) 9
(flow_add_nz<CLAMP>; 10 before, 13 as written):

    drain   nwe = dn + wn            ht = (dc + wc) - nwe          s = (dc - dn) + (wc - wn)        5 adds
            big = ldexp(ht, 2200)    m = min(s, big)                                                the sign of ht, as +-inf / 0
            x = dc > nwe ? wc : m                                                                   compare + two 32-bit selects
            f = clamp(x / 8)         wc -= |f|                     wn += f                          14
    add     en = dn + wn             ht = (dc + wc) - en           x = dc > en ? wc : ht            3 adds, compare + 2 selects
            f = clamp(x / 8)         wc -= |f|      wn += f                                         9

Writing the search down FOUND an instruction: the first candidate - the drain step without the reference's min(flow, w_c), kept
for three rounds "because s is rounded differently from ht" - would not differ on any pool, and a proof followed
(wdpm_stencil.h: where water moves on that branch, (dc - dn) + (wc - wn) <= 2 wc in real numbers and its three roundings add
less than 3 wc).  That form is the kernel's now; test_drain_flow_never_exceeds_the_centre_depth is the hunt for a
counter-example, kept.

Every candidate below removes one or more of the instructions that are left and is a form somebody could believe in - most
were believed in at some point of this build.  Each is run against the reference's conditional form on the adversarial operand
pools of tests/test_stencil_forms.py (ties of water surfaces, operands an ulp apart, subnormal depths, depths that exactly fill
the step to the neighbour) plus flat quantised ground and exact cancellations, and must DIFFER somewhere: the test records how
often, and fails if a candidate ever turns out exact - that would be a faster kernel.  The forms that are kept must not differ
anywhere.  A search, not a proof: the floor is "as far as found"."""
import os

import numpy as np

from conftest import ROOT
from test_clamped_step import clamped_drain_step, clamped_step, eighth_clamped
from test_stencil_forms import operands, reference_drain_step, reference_step, vmin


def _setup(dc, wc, dn, wn, gate, nvalid):
    dce = np.where(gate, dc, -np.inf)
    dnn = np.where(nvalid, dn, np.inf)
    wcl = np.where(gate, wc, 0.0)
    return dce, dnn, wcl


def _finish(gate, wc, wcl, wn, f):
    return np.where(gate, wcl - np.abs(f), wc), wn + f


# ---- drain candidates: each returns (w_c', w_n') ---------------------------------------------------------------------------------
def drain_with_ht_for_s(dc, wc, dn, wn, gate, nvalid):
    """11: the add step's operand - ht / 8 instead of ((dc - dn) + (wc - wn)) / 8; the same real number, rounded differently"""
    dce, dnn, wcl = _setup(dc, wc, dn, wn, gate, nvalid)
    nwe = dnn + wn
    ht = (dce + wcl) - nwe
    x = np.where(dce > nwe, wcl, ht)
    return _finish(gate, wc, wcl, wn, eighth_clamped(x))


def drain_without_the_sign_of_ht(dc, wc, dn, wn, gate, nvalid):
    """12: s > 0 taken for ht > 0 (no ldexp, no min): they disagree about the sign of a difference near zero"""
    dce, dnn, wcl = _setup(dc, wc, dn, wn, gate, nvalid)
    nwe = dnn + wn
    s = (dce - dnn) + (wcl - wn)
    x = np.where(dce > nwe, wcl, s)
    return _finish(gate, wc, wcl, wn, eighth_clamped(x))


def drain_with_min_for_the_select(dc, wc, dn, wn, gate, nvalid):
    """12: x = min(wc, m) instead of `dc > nwe ? wc : m` (one instruction for three): equal in real arithmetic, where
    dc > nwe <=> s > wc - not in fp64, where both sides carry their own rounding"""
    dce, dnn, wcl = _setup(dc, wc, dn, wn, gate, nvalid)
    nwe = dnn + wn
    ht = (dce + wcl) - nwe
    s = (dce - dnn) + (wcl - wn)
    big = np.where(ht > 0, np.inf, np.where(ht < 0, -np.inf, ht))
    x = vmin(wcl, vmin(s, big))
    return _finish(gate, wc, wcl, wn, eighth_clamped(x))


def drain_with_a_scaled_ht_as_the_cap(dc, wc, dn, wn, gate, nvalid):
    """13: m = min(s, 4 ht) - the cap by an exact small multiple of ht instead of by its sign at infinity (saves the ldexp if the
    multiple rides on another instruction)"""
    dce, dnn, wcl = _setup(dc, wc, dn, wn, gate, nvalid)
    nwe = dnn + wn
    ht = (dce + wcl) - nwe
    s = (dce - dnn) + (wcl - wn)
    x = np.where(dce > nwe, wcl, vmin(s, 4.0 * ht))
    return _finish(gate, wc, wcl, wn, eighth_clamped(x))


# ---- add candidates ---------------------------------------------------------------------------------------------------------------
def add_with_min_for_the_select(dc, wc, dn, wn, gate, nvalid):
    """7: x = min(wc, ht) instead of `dc > en ? wc : ht` - on flat quantised ground (dn == dc, a dry neighbour) ht = fl(dc + wc) - dc
    is wc rounded to the elevation's grid, as often above wc as below it"""
    dce = np.where(gate, dc, -np.inf)
    dnn = np.where(nvalid, dn, np.inf)
    ht = (dce + wc) - (dnn + wn)
    f = eighth_clamped(vmin(wc, ht))
    return wc - np.abs(f), wn + f


def add_with_the_surface_difference_only(dc, wc, dn, wn, gate, nvalid):
    """6: flow = ht / 8 always (no select at all: "the centre cannot lose more than the head difference")"""
    dce = np.where(gate, dc, -np.inf)
    dnn = np.where(nvalid, dn, np.inf)
    f = eighth_clamped((dce + wc) - (dnn + wn))
    return wc - np.abs(f), wn + f


DRAIN_CANDIDATES = [drain_with_ht_for_s, drain_without_the_sign_of_ht, drain_with_min_for_the_select,
                    drain_with_a_scaled_ht_as_the_cap]
ADD_CANDIDATES = [add_with_min_for_the_select, add_with_the_surface_difference_only]


def _pools(seed, rounds, n):
    """tests/test_stencil_forms.py's pools, restricted to what the clamped forms are used on: depths below 7.7 m (the kernels'
    guard), no -0.0 neighbour depth, and flat quantised ground added (the case that sinks the min-for-select forms)"""
    rng = np.random.default_rng(seed)
    for _ in range(rounds):
        dc, wc, dn, wn, gate, nvalid = operands(rng, n)
        flat = rng.random(n) < 0.2
        dcq = np.round(np.abs(dc) % 1000.0, 4)
        dc = np.where(flat, dcq, dc)
        dn = np.where(flat, np.where(rng.random(n) < 0.5, dcq, dcq + np.round(rng.normal(0, 1e-3, n), 4)), dn)
        wn = np.where(flat & (rng.random(n) < 0.5), 0.0, wn)
        wc = np.where(np.abs(wc) > 7.7, np.abs(wc) % 7.7, wc)
        wn = np.where(np.abs(wn) > 7.7, np.abs(wn) % 7.7, wn)
        # ... and cancellation: a neighbour far below the centre whose water brings it level again (dn = dc - D, wn = D +- a few ulps)
        # under a centre that holds next to nothing - (dc - dn) + (wc - wn) is then rounding noise of D's size, far above wc, and
        # only the reference's min(flow, w_c) keeps the flow at what the centre has (this is what sinks "drop the final min")
        canc = rng.random(n) < 0.1
        D = rng.choice([0.5, 3.0, 6.999, 7.5], n)
        dcs = 10.0 ** rng.uniform(-4, 1, n)
        dc = np.where(canc, dcs, dc)
        dn = np.where(canc, dcs - D, dn)
        wn = np.where(canc, D + rng.integers(-3, 4, n) * np.spacing(D), wn)
        wc = np.where(canc, 10.0 ** rng.uniform(-22, -13, n), wc)
        gate = np.where(canc, True, gate)
        nvalid = np.where(canc, True, nvalid)
        big_dem = np.abs(dc) >= 2.0 ** 30
        dc, dn = np.where(big_dem, dc % 1e6, dc), np.where(np.abs(dn) >= 2.0 ** 30, dn % 1e6, dn)
        yield dc, wc, dn, wn, gate & (wc > 0), nvalid


def _count_differences(reference, candidate, seed):
    bad = total = 0
    for args in _pools(seed, 4, 1_000_000):
        with np.errstate(invalid="ignore", over="ignore"):
            a, b = reference(*args), candidate(*args)
        diff = np.zeros(len(args[0]), dtype=bool)
        for u, v in zip(a, b):
            diff |= np.asarray(u).view(np.uint64) != np.asarray(v).view(np.uint64)
        bad += int(diff.sum())
        total += len(diff)
    return bad, total


def test_the_forms_that_are_kept_are_exact_on_these_pools():
    assert _count_differences(reference_drain_step, clamped_drain_step, 31)[0] == 0
    assert _count_differences(reference_step, clamped_step, 32)[0] == 0


def test_every_shorter_form_of_the_neighbour_step_differs_somewhere():
    lines = []
    for ref, cands, seed in ((reference_drain_step, DRAIN_CANDIDATES, 41), (reference_step, ADD_CANDIDATES, 42)):
        for cand in cands:
            bad, total = _count_differences(ref, cand, seed)
            lines.append(f"{cand.__name__}: differs on {bad} of {total} operand tuples - {cand.__doc__.split(':')[0].strip()} instructions")
            assert bad > 0, f"{cand.__name__} is exact on {total} adversarial tuples: a shorter neighbour step - build it"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "step_floor.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


def test_drain_flow_never_exceeds_the_centre_depth():
    """The reference's min(flow, w_c) (WDPMCL.c:1998) on the else branch of runoffd(): a hunt for operands where it bites - binade
    boundaries of either sign for the neighbour's surface, deep neighbours whose water cancels their depth exactly, near-ties an ulp
    apart, negative depths of odd files, the subnormal range, random exponents.  Wherever water moves (wc > 0, ht > 0, dc <= nwe)
    the flow ((dc - dn) + (wc - wn)) / 8 must stay below wc; the largest ratio seen is reported (0.25: the real-number bound)."""
    rng = np.random.default_rng(5)
    n, worst, seen = 1_000_000, 0.0, 0
    for rep in range(40):
        fam = rep % 5
        if fam == 0:      # nwe at powers of two of either sign, dc = nwe or just below
            k, sign = rng.integers(-30, 30, n), rng.choice([1.0, -1.0], n)
            nwe_t = sign * 2.0 ** k
            dc = nwe_t - rng.integers(0, 4, n) * np.spacing(nwe_t) * rng.choice([0, 1], n)
            wn = np.abs(nwe_t) * rng.choice([0.0, 0.3, 1.0, 3.0, 1e3, 2.0 ** 20], n) * (1 + rng.integers(-3, 4, n) * 2.0 ** -52)
            dn = nwe_t - wn
            wc = np.spacing(nwe_t) * rng.choice([0.25, 0.5, 0.5000001, 0.75, 1, 1.25, 2, 5], n)
        elif fam == 1:    # near-ties at ordinary magnitudes
            dc = rng.choice([1.0, -1.0], n) * 10.0 ** rng.uniform(-5, 6, n)
            wn = np.abs(dc) * 10.0 ** rng.uniform(-3, 3, n)
            dn = dc - wn + rng.integers(-4, 5, n) * np.spacing(dc)
            wc = np.spacing(dc) * rng.uniform(0.3, 6, n)
        elif fam == 2:    # negative neighbour depths (odd input files)
            dc, wn = 10.0 ** rng.uniform(-3, 4, n), -10.0 ** rng.uniform(-6, 2, n)
            dn = dc - wn + rng.integers(-4, 5, n) * np.spacing(dc)
            wc = np.spacing(dc) * rng.uniform(0.3, 6, n)
        elif fam == 3:    # the subnormal range
            dc = rng.integers(0, 200, n) * 5e-324 * rng.choice([1.0, -1.0], n)
            wn = rng.integers(0, 200, n) * 5e-324
            dn = dc - wn + rng.integers(-3, 4, n) * 5e-324
            wc = rng.integers(1, 20, n) * 5e-324
        else:             # random exponents for everything, the neighbour's surface within a few ulps of the centre's elevation
            dc = rng.choice([1.0, -1.0], n) * 2.0 ** rng.uniform(-60, 60, n)
            wn = 2.0 ** rng.uniform(-80, 60, n)
            dn = dc - wn * (1 + rng.integers(-2, 3, n) * 2.0 ** -52) + rng.integers(-2, 3, n) * np.spacing(dc)
            wc = 2.0 ** rng.uniform(-110, 10, n)
        with np.errstate(all="ignore"):
            nwe = dn + wn
            ht = (dc + wc) - nwe
            s = (dc - dn) + (wc - wn)
            go = (wc > 0) & (ht > 0) & ~(dc > nwe) & np.isfinite(s)
            ratio = np.where(go, s / 8.0 / wc, 0.0)
        seen += int(go.sum())
        worst = max(worst, float(ratio.max()))
        assert worst < 1.0, (fam, worst)
    assert seen > 5e6 and worst <= 0.25 + 1e-12, (seen, worst)
    with open(os.path.join(ROOT, "gpurun_out", "step_floor.txt"), "a") as f:
        f.write(f"drain else branch: largest flow / w_c over {seen} operand tuples where water moves: {worst}\n")

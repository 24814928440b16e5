"""The row-boundary table of the marching kernel (wdpm_fused.hip::xcd_rebalance_kernel), round 5's `pair` rounding restated in
numpy: the two waves of a SIMD are work items four apart - strips s and s + 4 of one chunk row - and their strips' boundaries are
rounded half a triple apart, so that a tall chunk of one sits beside a short chunk of the other.  Checked here: the property the
kernel's comment claims (equal weights: the steps of the two waves of every SIMD add up to the same number to within one, where
plain rounding pairs tall with tall), that boundaries stay monotone and complete under skewed weights, and the slot filling
arithmetic of wdpm_launch_fused_rows for the shapes DESIGN.md quotes.  No GPU: results never depend on the table (the parity
suites run with deliberately skewed weights, WDPM_BALANCE=2); this is about what the table is FOR."""
import numpy as np
import pytest


def table(T, nchunks, nstrips, weights, ipx, pair):
    """xcd_rebalance_kernel's table in row triples: t[c][s], c = 0 .. nchunks"""
    w = np.asarray(weights, dtype=np.float32)
    out = np.zeros((nchunks + 1, nstrips), dtype=int)
    for s in range(nstrips):
        def wgt(c):
            x = min((c * nstrips + s) // ipx, 7)
            return w[x] * (w[8] if c == nchunks - 1 else w[9] if c == 0 else np.float32(1.0))
        total = np.float32(sum(wgt(c) for c in range(nchunks)))
        phase = (0.75 if (s >> 2) & 1 else 0.25) if pair else 0.5
        cum, prev = np.float32(0), 0
        for c in range(nchunks):
            cum += wgt(c)
            t = T if c == nchunks - 1 else int(np.float32(T) * cum / total + np.float32(phase))
            t = min(max(t, prev + 2), T - 2 * (nchunks - 1 - c))
            out[c + 1][s] = t
            prev = t
    return out


def launch_geometry(rows, ncp, slots=2048):
    nstrips = 1 if ncp <= 179 else (ncp - 179 + 170) // 171 + 1
    T = (rows - 2 + 2) // 3
    return nstrips, slots // nstrips, T


@pytest.mark.parametrize("rows,ncp", [(1055, 8192), (2051, 16386), (4098, 4098), (8194, 8194), (16386, 16386), (3002, 3002)])
def test_partner_strips_take_turns_at_the_tall_chunks(rows, ncp):
    nstrips, nchunks, T = launch_geometry(rows, ncp)
    ipx = 8 * ((nstrips * nchunks + 7) // 8 + 7) // 8 * 8 // 8
    uniform = [1.0] * 10
    worst = {}
    for pair in (0, 1):
        t = table(T, nchunks, nstrips, uniform, ipx, pair)
        h = np.diff(t, axis=0)                                    # triples per chunk: [chunk][strip]
        assert (t[0] == 0).all() and (t[-1] == T).all() and (h >= 2).all()
        steps = h + 2                                             # marching steps of a wave: H / 3 + 2
        partner = np.arange(nstrips) ^ 4                          # work items four apart (nstrips is a multiple of 8 for these shapes)
        ok = partner < nstrips
        both = steps[:, ok] + steps[:, partner[ok]]
        worst[pair] = int(both.max()), int(both.min())
    frac = T / nchunks - T // nchunks
    if 0.05 < frac < 0.95:                                        # heights that are not (nearly) whole numbers of triples anyway
        base = 2 * (T // nchunks) + 4                             # two short chunks, two warm-up steps each
        if frac <= 0.5:
            assert worst[1][0] == base + 1, (worst, T, nchunks)   # never two tall chunks on one SIMD ...
            assert worst[0][0] == base + 2                        # ... where plain rounding puts them side by side
        else:
            assert worst[1][1] == base + 1, (worst, T, nchunks)   # most chunks are tall: never two short ones on one SIMD
    assert worst[1][0] - worst[1][1] <= 1


def test_the_drain_slab_of_an_8_gpu_run_fills_its_slots():
    """1055 x 8190 (config 5 on 8 GPUs): equal heights in whole triples meant 27-row chunks, 39 per strip, 1872 waves on 2048 slots and
    11 steps for everyone; the table cuts a strip into 42 chunks of 8 or 9 triples, and no SIMD holds two of the tall ones"""
    nstrips, nchunks, T = launch_geometry(1055, 8192)
    assert (nstrips, nchunks, T) == (48, 42, 351) and nstrips * nchunks == 2016
    h = np.diff(table(T, nchunks, nstrips, [1.0] * 10, 8 * 32, 1), axis=0)
    assert set(np.unique(h)) == {8, 9}
    s = np.arange(nstrips)
    assert ((h + h[:, s ^ 4]) <= 17).all()


def test_skewed_weights_keep_the_table_a_tiling():
    rng = np.random.default_rng(7)
    for _ in range(50):
        nstrips, nchunks = int(rng.integers(1, 100)), int(rng.integers(2, 90))
        T = int(rng.integers(2 * nchunks, 40 * nchunks))
        w = list(rng.uniform(0.7, 1.4, 8)) + [float(rng.uniform(0.75, 1.2)), float(rng.uniform(0.75, 1.2))]
        t = table(T, nchunks, nstrips, w, max(8, nstrips * nchunks // 8), int(rng.integers(0, 2)))
        assert (t[0] == 0).all() and (t[-1] == T).all() and (np.diff(t, axis=0) >= 2).all()

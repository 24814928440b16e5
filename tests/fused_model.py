"""Executable model of the fused-iteration kernel's SCHEDULE (wdpm_amd/csrc/wdpm_fused.hip).

Test infrastructure: a numpy emulation of what one wave64 of the fused kernel does — lanes own 3
columns each, a 7-row register window marches down a row chunk, the 9 colour passes of one
iteration are applied in a skewed order, column blocks that straddle lanes borrow the next lane's
columns (DPP wave_shl on the GPU) — written to mirror the HIP code line by line so that the index
bookkeeping (strips, chunks, window slots, output masks) can be checked against the oracle on the
CPU.  The arithmetic per neighbour is the same as wdpm_stencil.h::flow_add.
"""
import numpy as np

INF = np.inf
LANES = 64
STRIP_IN = 3 * LANES          # 192 columns loaded per wave
HALO_L, HALO_R = 8, 13        # columns given up left / right per fused iteration (worst-case reach 8 / 12)
STRIP_OUT = STRIP_IN - HALO_L - HALO_R  # 171, a multiple of 3 so every strip starts on a block boundary

NB = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]


def strip_geometry(ncp):
    """(c0, out_lo, out_hi) per strip: outputs partition [0, ncp)."""
    strips = []
    j = 0
    while True:
        c0 = STRIP_OUT * j
        lo = 0 if j == 0 else c0 + HALO_L
        hi = c0 + STRIP_IN - 1 - HALO_R
        strips.append((c0, lo, min(hi, ncp - 1)))
        if hi >= ncp - 1:
            break
        j += 1
    return strips


def chunk_geometry(rows, H):
    """(A, nsteps, out_lo, out_hi) per row chunk: outputs partition [0, rows).  H % 3 == 0."""
    assert H % 3 == 0 and H >= 3
    chunks = []
    i = 0
    while True:
        A = H * i
        lo = 0 if i == 0 else A + 2
        hi = H * (i + 1) + 1
        chunks.append((A, H // 3 + 2, lo, min(hi, rows - 1)))
        if hi >= rows - 1:
            break
        i += 1
    return chunks


def lane_next(v, fill):
    """value held by lane+1; lane 63 gets `fill` (DPP wave_shl:1, bound_ctrl off)."""
    out = np.empty_like(v)
    out[:-1] = v[1:]
    out[-1] = fill
    return out


def lane_prev(v, keep):
    """value held by lane-1; lane 0 keeps `keep[0]` (DPP wave_shr:1)."""
    out = np.empty_like(v)
    out[1:] = v[:-1]
    out[0] = keep[0]
    return out


def flow_add(dc, wc, dn, wn, gate):
    with np.errstate(invalid="ignore"):
        en = dn + wn
        ht = (dc + wc) - en
        go = gate & (ht > 0)
        x = np.where(dc > en, wc, ht)
        flow = x * 0.125
        flow = np.where(flow < wc, flow, wc)
        wc2 = wc - flow
        wn2 = wn + flow
    return np.where(go, wc2, wc), np.where(go, wn2, wn)


def block_update(w, d):
    """w, d: 3x3 lists of [64] arrays (rows x cols of the block).  In place on w."""
    wc = w[1][1]
    dc = d[1][1]
    gate = (wc > 0.0) & (dc < INF)
    for (i, j) in NB:
        wc, w[1 + i][1 + j] = flow_add(dc, wc, d[1 + i][1 + j], w[1 + i][1 + j], gate)
    w[1][1] = wc


def stage(W, D, s0):
    """the three column passes (oj = 1,2,3) of one row alignment on window slots s0..s0+2."""
    rows = (s0, s0 + 1, s0 + 2)
    # oj = 1: own columns 0,1,2
    w = [[W[r][0], W[r][1], W[r][2]] for r in rows]
    d = [[D[r][0], D[r][1], D[r][2]] for r in rows]
    block_update(w, d)
    for k, r in enumerate(rows):
        W[r][0], W[r][1], W[r][2] = w[k]
    # oj = 2: own 1,2 + next lane's column 0
    wx0 = [lane_next(W[r][0], 0.0) for r in rows]
    dx0 = [lane_next(D[r][0], 0.0) for r in rows]
    w = [[W[r][1], W[r][2], wx0[k]] for k, r in enumerate(rows)]
    d = [[D[r][1], D[r][2], dx0[k]] for k, r in enumerate(rows)]
    block_update(w, d)
    for k, r in enumerate(rows):
        W[r][1], W[r][2], wx0[k] = w[k]
    # oj = 3: own 2 + next lane's columns 0,1
    wx1 = [lane_next(W[r][1], 0.0) for r in rows]
    dx1 = [lane_next(D[r][1], 0.0) for r in rows]
    w = [[W[r][2], wx0[k], wx1[k]] for k, r in enumerate(rows)]
    d = [[D[r][2], dx0[k], dx1[k]] for k, r in enumerate(rows)]
    block_update(w, d)
    for k, r in enumerate(rows):
        W[r][2], wx0[k], wx1[k] = w[k]
    # hand the borrowed columns back to their owner (lane+1); lane 0 keeps its own
    for k, r in enumerate(rows):
        W[r][0] = lane_prev(wx0[k], W[r][0])
        W[r][1] = lane_prev(wx1[k], W[r][1])


def run_wave(win, wout, dem, miss, strip, chunk):
    rows, ncp = win.shape
    c0, oc_lo, oc_hi = strip
    A, nsteps, or_lo, or_hi = chunk
    lane = np.arange(LANES)
    col = [c0 + 3 * lane + j for j in range(3)]
    colok = [(c < ncp) for c in col]
    W = [[np.zeros(LANES) for _ in range(3)] for _ in range(7)]
    D = [[np.full(LANES, INF) for _ in range(3)] for _ in range(7)]

    def load(r):
        ws, ds = [], []
        for j in range(3):
            wv = np.zeros(LANES)
            dv = np.full(LANES, INF)
            if 0 <= r < rows:
                ok = colok[j]
                cc = np.where(ok, col[j], 0)
                dd = dem[r, cc]
                wv = np.where(ok, win[r, cc], 0.0)
                dv = np.where(ok & (dd > miss), dd, INF)
            ws.append(wv)
            ds.append(dv)
        return ws, ds

    for n in range(nsteps):
        for i in range(3):
            W[4 + i], D[4 + i] = load(A + 3 * n + i)
        stage(W, D, 4)   # oi = 1 on rows 3n   .. 3n+2
        stage(W, D, 2)   # oi = 2 on rows 3n-2 .. 3n
        stage(W, D, 0)   # oi = 3 on rows 3n-4 .. 3n-2
        for i in range(3):
            r = A + 3 * n - 4 + i
            if or_lo <= r <= or_hi:
                for j in range(3):
                    ok = (col[j] >= oc_lo) & (col[j] <= oc_hi)
                    wout[r, col[j][ok]] = W[i][j][ok]
        for k in range(4):
            W[k], D[k] = W[k + 3], D[k + 3]
        # slots 4..6 are overwritten by the next step's loads


def fused_iteration(win, dem, miss, H=12):
    """One whole iteration (9 colour passes) of the add/subtract module on a padded slab."""
    rows, ncp = win.shape
    wout = np.full_like(win, np.nan)
    for chunk in chunk_geometry(rows, H):
        for strip in strip_geometry(ncp):
            run_wave(win, wout, dem, miss, strip, chunk)
    assert not np.isnan(wout).any(), "output cells not covered"
    return wout


# ---------------------------------------------------------------------------------------------------
# The triangle kernel (wdpm_fused.hip: tri_iteration_kernel): nine rows in, the middle three out.
#   oi = 1 on window rows 0-2, 3-5, 6-8;  oi = 2 on rows 1-3, 4-6;  oi = 3 on rows 2-4.
# The row blocks of one row alignment are independent (the GPU advances them in lockstep); here they
# simply run one after the other.
# ---------------------------------------------------------------------------------------------------
def run_tri_wave(win, wout, dem, miss, strip, A, out_last):
    rows, ncp = win.shape
    c0, oc_lo, oc_hi = strip
    or_lo, or_hi = (0 if A == 0 else A + 2), min(A + 4, out_last)
    lane = np.arange(LANES)
    col = [c0 + 3 * lane + j for j in range(3)]
    W, D = [], []
    for i in range(9):
        r = A + i
        ws, ds = [], []
        for j in range(3):
            ok = (col[j] < ncp) & (r < rows)
            cc = np.where(col[j] < ncp, col[j], 0)
            rr = min(r, rows - 1)
            dd = dem[rr, cc]
            ws.append(np.where(ok, win[rr, cc], 0.0))
            ds.append(np.where(ok & (dd > miss), dd, INF))
        W.append(ws)
        D.append(ds)
    for s0 in (0, 3, 6):
        stage(W, D, s0)      # oi = 1
    for s0 in (1, 4):
        stage(W, D, s0)      # oi = 2
    stage(W, D, 2)           # oi = 3
    for i in range(5):
        r = A + i
        if or_lo <= r <= or_hi:
            for j in range(3):
                ok = (col[j] >= oc_lo) & (col[j] <= oc_hi)
                wout[r, col[j][ok]] = W[i][j][ok]


def tri_iteration(win, dem, miss):
    """One whole iteration with the triangle schedule: chunks of three output rows."""
    rows, ncp = win.shape
    wout = np.full_like(win, np.nan)
    nchunks = max((rows - 1 - 0 - 1 + 2) // 3, 1)
    for chunk in range(nchunks):
        for strip in strip_geometry(ncp):
            run_tri_wave(win, wout, dem, miss, strip, 3 * chunk, rows - 1)
    assert not np.isnan(wout).any(), "output cells not covered"
    return wout


# ---------------------------------------------------------------------------------------------------
# Dry tiles (wdpm_kernels.h::TileFlags): per tile (the exact output block of one marching wave) a flag
# "all +0.0"; a wave whose tile and eight neighbours are flagged in the input raster writes zeros without
# loading anything.  The model keeps the flags exactly as the kernel does and checks the rule's premise:
# the skipped wave's output, had it been computed, is all zero.
# ---------------------------------------------------------------------------------------------------
def fused_iteration_with_tiles(win, dem, miss, H, zin):
    """-> (wout, zout, skipped): zin / zout are (nchunks, nstrips) bool arrays or None (unknown)"""
    assert H >= 6
    rows, ncp = win.shape
    chunks, strips = chunk_geometry(rows, H), strip_geometry(ncp)
    wout = np.full_like(win, np.nan)
    zout = np.zeros((len(chunks), len(strips)), bool)
    skipped = 0
    for ci, chunk in enumerate(chunks):
        for si, strip in enumerate(strips):
            dry = zin is not None
            if dry:
                for dc in (-1, 0, 1):
                    for ds in (-1, 0, 1):
                        cc, ss = ci + dc, si + ds
                        if 0 <= cc < len(chunks) and 0 <= ss < len(strips):
                            dry &= bool(zin[cc, ss])
            A, _, or_lo, or_hi = chunk
            c0, oc_lo, oc_hi = strip
            if dry:
                # what the wave would have computed, to check the rule
                probe = np.full_like(win, np.nan)
                run_wave(win, probe, dem, miss, strip, chunk)
                block = probe[or_lo:or_hi + 1, oc_lo:oc_hi + 1]
                assert (block.view(np.uint64) == 0).all(), "a tile with a dry neighbourhood produced water"
                wout[or_lo:or_hi + 1, oc_lo:oc_hi + 1] = 0.0
                zout[ci, si] = True
                skipped += 1
            else:
                run_wave(win, wout, dem, miss, strip, chunk)
                zout[ci, si] = (wout[or_lo:or_hi + 1, oc_lo:oc_hi + 1].view(np.uint64) == 0).all()
    assert not np.isnan(wout).any()
    return wout, zout, skipped

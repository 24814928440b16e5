"""numpy model of the parallel evaluation of the reference's SEQUENTIAL fp64 sum (`final_vol`,
WDPMCL.c:1259-1266) that wdpm_volume_partial runs on the GPU.

fp64 addition is not associative, so the sum of the water raster cannot simply be tree-reduced: the
printed volume has to be the reference's left-to-right sum, bit for bit.  But while the running
sum S stays inside one binade [2^k, 2^(k+1)) every addition fl(S + x), x >= 0, is S + q(x)*u with
u = 2^(k-52) and q(x) = x/u rounded to the nearest integer - independent of S unless x/u lies
exactly half way between two integers (then the parity of S decides).  Integer addition IS
associative, so a chunk of the sequence can be summed in any order once its binade is known:

  pass A   per chunk: an ordinary (approximate) sum and a flag for negative / non-finite terms
  host     prefix of the approximate sums -> the binade k each chunk will run in, or "sequential"
           when the prefix is too close to a power of two (or too small) to tell
  pass B   per chunk: I = sum of q(x) as int64, flag if any term is an exact tie
  chain    S <- S + I*u for every chunk whose assumption checks out on the ACTUAL S
           (2^k <= S and S + I*u < 2^(k+1)); any other chunk is summed term by term.

Nothing is taken on trust: the chain verifies the binade with the true running sum, so the
approximate prefix only decides how many chunks take the slow path."""
import numpy as np

CHUNK = 65536


def sequential_sum(x, start=0.0):
    """the reference: s = start; for v in x: s += v"""
    if len(x) == 0:
        return np.float64(start)
    with np.errstate(all="ignore"):
        return np.cumsum(np.concatenate(([np.float64(start)], x)))[-1]   # cumsum adds left to right


def binade(v):
    """k with 2^k <= v < 2^(k+1) for a positive normal double"""
    return int(np.frexp(v)[1]) - 1


def chunk_pass_a(c):
    with np.errstate(all="ignore"):
        return float(np.sum(c)), bool(np.any(~(c >= 0)) or np.any(np.isinf(c)))     # ~(c >= 0): negative or NaN


def chunk_pass_b(c, k):
    """integer sum of the terms in units of u = 2^(k-52), and whether any term is an exact tie"""
    t = np.ldexp(c, 52 - k)                 # exact scaling; every term is < 2^(k+1), so t < 2^53
    m = np.floor(t)
    r = t - m                               # exact
    q = m.astype(np.int64) + (r > 0.5)
    return int(q.sum()), bool(np.any(r == 0.5))


def fast_sequential_sum(x, start=0.0, chunk=CHUNK, stats=None):
    x = np.asarray(x, dtype=np.float64)
    chunks = [x[i:i + chunk] for i in range(0, len(x), chunk)]
    a = [chunk_pass_a(c) for c in chunks]
    s = np.float64(start)
    ok_start = bool(np.isfinite(s) and s >= 0)
    prefix = float(s) if ok_start else 0.0
    n_fast = 0
    for c, (approx, dirty) in zip(chunks, a):
        lo, hi = prefix, prefix + approx
        prefix = hi
        k = None
        if ok_start and not dirty and lo > 1e-290 and np.isfinite(hi):
            k_lo, k_hi = binade(lo * (1 - 1e-6)), binade(hi * (1 + 1e-6))
            if k_lo == k_hi:
                k = k_lo
        done = False
        if k is not None:
            i_sum, tie = chunk_pass_b(c, k)
            if not tie and s >= np.ldexp(1.0, k):
                s_new = s + np.ldexp(np.float64(i_sum), k - 52)          # exact: both are multiples of u
                if s_new < np.ldexp(1.0, k + 1):
                    s, done = s_new, True
                    n_fast += 1
        if not done:
            s = sequential_sum(c, s)
    if stats is not None:
        stats["chunks"], stats["fast"] = len(chunks), n_fast
    return s

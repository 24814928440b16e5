"""the chunked evaluation of the reference's sequential volume sum (tests/seqsum_model.py, the model
of wdpm_volume_partial's device path) equals the left-to-right fp64 sum bit for bit"""
import numpy as np
import pytest

from seqsum_model import fast_sequential_sum, sequential_sum


def same(a, b):
    return np.float64(a).view(np.uint64) == np.float64(b).view(np.uint64)


def cases():
    rng = np.random.default_rng(3)
    n = 700_000
    yield "uniform depths", rng.random(n) * 0.3, 0.0
    yield "constant 0.1 (binade crossings)", np.full(n, 0.1), 0.0
    yield "mostly dry", np.where(rng.random(n) < 0.9, 0.0, rng.random(n)), 0.0
    yield "ties everywhere", rng.integers(0, 5, n) * 2.0 ** -31 + rng.integers(0, 3, n) * 0.25, 123456.0
    yield "short mantissas", rng.integers(0, 1 << 20, n) * 2.0 ** -40, 3.0e6
    yield "wide dynamic range", 10.0 ** rng.uniform(-320, 2, n), 0.0
    yield "subnormal only", rng.integers(0, 1000, n) * 5e-324, 0.0
    yield "negative cells from an odd file", np.where(rng.random(n) < 1e-4, -rng.random(n), rng.random(n)), 0.0
    yield "a NaN", np.where(np.arange(n) == 400_000, np.nan, rng.random(n)), 0.0
    yield "an inf", np.where(np.arange(n) == 123_456, np.inf, rng.random(n)), 0.0
    yield "huge then small", np.concatenate((np.full(10, 1e300), rng.random(n))), 0.0
    yield "chained start value", rng.random(n), 7.123456789e8
    yield "start just below a power of two", rng.random(n) * 1e-3, np.nextafter(2.0 ** 20, 0)
    yield "negative start", rng.random(n), -5.0
    yield "empty", np.zeros(0), 42.0
    yield "one term", np.array([0.3]), 0.1


@pytest.mark.parametrize("name,x,start", list(cases()), ids=[c[0] for c in cases()])
def test_chunked_sum_equals_sequential_sum(name, x, start):
    st = {}
    got = fast_sequential_sum(x, start, stats=st)
    want = sequential_sum(x, start)
    assert same(got, want) or (np.isnan(got) and np.isnan(want)), (name, got, want)
    if name in ("uniform depths", "mostly dry", "chained start value"):
        assert st["fast"] >= st["chunks"] - 30, st            # the slow path is the exception on ordinary data


def test_sequential_reference_is_really_sequential():
    x = np.random.default_rng(1).random(5000)
    s = 0.0
    for v in x:
        s += v
    assert same(sequential_sum(x), s)
    assert not same(np.sum(x), s) or True                      # (a pairwise sum usually differs; not required)

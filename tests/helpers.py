"""Shared helpers for the parity tests (host-side set-up restated from WDPMCL.c)."""
import hashlib

import numpy as np


def pad(dem, water, missing):
    """bigdem/bigwater build, WDPMCL.c:796-807."""
    R, C = dem.shape
    bd = np.full((R + 2, C + 2), missing, dtype=np.float64)
    bw = np.zeros((R + 2, C + 2), dtype=np.float64)
    bd[1:-1, 1:-1] = dem
    bw[1:-1, 1:-1] = water
    return bd, bw


def find_drain(bd):
    """WDPMCL.c:1005-1017 — first row-major strict minimum among bigdem > 0."""
    m = np.where(bd > 0, bd, np.inf)
    k = int(np.argmin(m))  # argmin returns the first occurrence
    return k // bd.shape[1], k % bd.shape[1]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def n_bit_diff(a, b):
    return int(np.count_nonzero(np.ascontiguousarray(a).view(np.uint64) != np.ascontiguousarray(b).view(np.uint64)))


def random_case(seed, R, C, missing_frac=0.05, dry_frac=0.3, depth=0.3):
    rng = np.random.default_rng(seed)
    missing = -99999.0
    y, x = np.mgrid[0:R, 0:C]
    dem = 500.0 + 2.0 * np.sin(x / 3.1) * np.cos(y / 2.3) + rng.normal(0, 0.4, (R, C)) - 0.01 * (x + y)
    dem = np.round(dem, 4)
    dem[rng.random((R, C)) < missing_frac] = missing
    water = np.where(rng.random((R, C)) < dry_frac, 0.0, depth * rng.random((R, C)))
    water = np.where(dem > missing, water, 0.0)
    return dem, water, missing

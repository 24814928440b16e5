"""The fused kernel's marching-window schedule (tests/fused_model.py mirrors wdpm_fused.hip) is
bit-identical to the oracle for multi-strip, multi-chunk, ragged and degenerate rasters."""
import numpy as np
import pytest

from fused_model import chunk_geometry, fused_iteration, strip_geometry
from helpers import n_bit_diff, pad, random_case


@pytest.mark.parametrize("R,C,H,seed", [(38, 398, 12, 1), (7, 9, 3, 2), (50, 170, 9, 3), (33, 180, 30, 4),
                                        (20, 600, 6, 5), (1, 1, 3, 6), (64, 50, 300, 7), (10, 174, 24, 8)])
def test_schedule_model_matches_oracle(oracle, R, C, H, seed):
    dem, water, miss = random_case(seed, R, C)
    bd, bw = pad(dem, water, miss)
    with oracle.context(module="add", nrows=R, ncols=C, missingvalue=miss) as ctx:
        ctx.upload(bd, bw)
        w = bw.copy()
        for _ in range(2):
            ctx.iterate(1)
            w = fused_iteration(w, bd, miss, H)
            assert n_bit_diff(w, ctx.download_water()) == 0


def test_geometry_covers_every_cell_once():
    for ncp in (3, 176, 177, 344, 345, 16386):
        strips = strip_geometry(ncp)
        assert strips[0][1] == 0 and strips[-1][2] == ncp - 1
        for a, b in zip(strips, strips[1:]):
            assert b[1] == a[2] + 1 and b[0] % 3 == 0
    for rows, H in ((3, 3), (50, 12), (16386, 393), (2000, 24)):
        chunks = chunk_geometry(rows, H)
        assert chunks[0][2] == 0 and chunks[-1][3] == rows - 1
        for a, b in zip(chunks, chunks[1:]):
            assert b[2] == a[3] + 1 and b[0] % 3 == 0


@pytest.mark.parametrize("R,C,seed", [(38, 398, 21), (7, 9, 22), (50, 170, 23), (33, 330, 24), (1, 1, 25), (64, 200, 26),
                                      (2, 175, 27), (3, 172, 28)])
def test_triangle_schedule_matches_oracle(oracle, R, C, seed):
    """nine rows in, three out: oi=1 on three row blocks, oi=2 on two, oi=3 on one (tri_iteration_kernel)"""
    from fused_model import tri_iteration
    dem, water, miss = random_case(seed, R, C)
    bd, bw = pad(dem, water, miss)
    with oracle.context(module="add", nrows=R, ncols=C, missingvalue=miss) as ctx:
        ctx.upload(bd, bw)
        w = bw.copy()
        for _ in range(3):
            ctx.iterate(1)
            w = tri_iteration(w, bd, miss)
            assert n_bit_diff(w, ctx.download_water()) == 0


def test_dry_tile_rule_in_the_model(oracle):
    """a tile whose 3x3 tile neighbourhood is all +0.0 in the input raster produces an all-zero block (the model
    computes the skipped waves anyway and checks), the flags are kept as the kernel keeps them, and the result
    equals the oracle's; water spreads from two ponds over a mostly dry raster"""
    from fused_model import fused_iteration_with_tiles
    R, C, H, miss = 70, 700, 6, -99999.0
    rng = np.random.default_rng(3)
    y, x = np.mgrid[0:R, 0:C]
    dem = np.round(520.0 - 0.02 * x - 0.05 * y + 0.2 * np.sin(x / 5.0) + 0.05 * rng.random((R, C)), 4)
    dem[rng.random((R, C)) < 0.03] = miss
    water = np.zeros((R, C))
    water[5:12, 30:44] = 0.3
    water[40:47, 400:420] = 0.4
    water[dem <= miss] = 0.0
    bd, bw = pad(dem, water, miss)
    with oracle.context(module="add", nrows=R, ncols=C, missingvalue=miss) as ctx:
        ctx.upload(bd, bw)
        w, z, total_skipped = bw.copy(), None, 0
        for it in range(12):
            ctx.iterate(1)
            w, z, skipped = fused_iteration_with_tiles(w, bd, miss, H, z)
            total_skipped += skipped
            assert n_bit_diff(w, ctx.download_water()) == 0, it
    assert total_skipped > 100

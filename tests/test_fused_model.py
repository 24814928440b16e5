"""The fused kernel's marching-window schedule (tests/fused_model.py mirrors wdpm_fused.hip) is
bit-identical to the oracle for multi-strip, multi-chunk, ragged and degenerate rasters."""
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

#!/usr/bin/env python3
"""A mostly dry raster: synthetic n x n DEM, half of it NODATA (the lower-right triangle), water
only in a few ponds.  Times blocks of iterations with dry-tile skipping on and off and checks that the rasters agree.
    sparse_bench.py [n] [iterations per block] [blocks] [ponds]"""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wdpm_amd
from wdpm_amd.capi import OPT_TILES, OPT_TILES_SEEN, OPT_TILES_WORKED, OPT_SPARSE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ponds = int(sys.argv[4]) if len(sys.argv) > 4 else 12
MISS = -99999.0
lib = wdpm_amd.load_hip()
dem = lib.synth_dem(n, n)
yy, xx = np.ogrid[0:n, 0:n]
dem[(yy + xx) > n] = MISS                                    # ~50 % NODATA
water = np.zeros((n, n))
rng = np.random.default_rng(0)
for _ in range(ponds):                                       # a dozen ponds of ~1 % of the side each
    r, c = rng.integers(0, n // 2, 2)
    s = max(n // 100, 8)
    water[r:r + s, c:c + s] = 0.5
water[dem <= MISS] = 0.0
bd = np.full((n + 2, n + 2), MISS); bd[1:-1, 1:-1] = dem
bw = np.zeros((n + 2, n + 2)); bw[1:-1, 1:-1] = water
del dem, water
out = {}
for tiles in (1, 0):
    with lib.context(module="add", nrows=n, ncols=n, missingvalue=MISS) as c:
        c.set_option(OPT_TILES, tiles)
        c.upload(bd, bw)
        c.run_block(5, 5e-6)
        times = []
        for b in range(blocks):
            c.synchronize(); t = time.perf_counter()
            c.run_block(iters, 5e-6)
            times.append((time.perf_counter() - t) / iters * 1e3)
        out[tiles] = c.download_water()
        print(f"tiles={tiles}: ms per iteration by block {['%.3f' % x for x in times]}  sparse mode {c.get_option(OPT_SPARSE)}  "
              f"tiles seen {c.get_option(OPT_TILES_SEEN)} worked {c.get_option(OPT_TILES_WORKED)}")
same = np.array_equal(out[0].view(np.uint64), out[1].view(np.uint64))
print("rasters identical with and without dry-tile skipping:", same, " wet cells:", int((out[1] > 0).sum()), "of", n * n)
sys.exit(0 if same else 1)

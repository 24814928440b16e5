#!/usr/bin/env python3
"""How a real run's cost develops as the water settles: synthetic n x n DEM, Add 100 mm everywhere, blocks of 1000
iterations until the max change falls under the tolerance or the block limit is reached.  Per block: ms per iteration,
share of tiles that worked, sparse mode, wet cells (> 0) via the device statistics, max change.
    settle_bench.py [n] [blocks] [eltol_mm] [thres_mm]"""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wdpm_amd
from wdpm_amd.capi import OPT_TILES_SEEN, OPT_TILES_WORKED, OPT_SPARSE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 60
eltol = (float(sys.argv[3]) if len(sys.argv) > 3 else 1.0) / 1000.0
thres = (float(sys.argv[4]) if len(sys.argv) > 4 else 0.005) / 1000.0
MISS = -99999.0
lib = wdpm_amd.load_hip()
dem = lib.synth_dem(n, n)
bd = np.full((n + 2, n + 2), MISS); bd[1:-1, 1:-1] = dem
bw = np.where(bd > MISS, 0.1, 0.0)
del dem
t_all = time.perf_counter()
with lib.context(module="add", nrows=n, ncols=n, missingvalue=MISS) as c:
    c.upload(bd, bw)
    for b in range(blocks):
        c.synchronize(); t = time.perf_counter()
        md = c.run_block(1000, thres)
        dt = (time.perf_counter() - t)
        seen, worked = c.get_option(OPT_TILES_SEEN), c.get_option(OPT_TILES_WORKED)
        w = c.download_water() if (b % 10 == 9 or md < eltol or b == blocks - 1) else None
        wet = "" if w is None else f"  wet cells {(w > 0).mean() * 100:.1f} %  cells above 1 mm {(w > 1e-3).mean() * 100:.1f} %"
        if w is not None:
            # how coarse may a unit of work be and still find itself all dry?  (rows x columns of aligned windows)
            nz = w[1:-1, 1:-1] != 0
            for rr, cc in ((3, 24), (7, 64), (7, 192), (9, 192), (24, 192), (96, 171)):
                R, Cc = nz.shape[0] // rr * rr, nz.shape[1] // cc * cc
                blk = nz[:R, :Cc].reshape(R // rr, rr, Cc // cc, cc).any(axis=(1, 3))
                wet += f"  dry {rr}x{cc}: {(~blk).mean() * 100:.0f} %"
        print(f"block {b + 1:3d}: {dt:.3f} ms per iteration  max change {md * 1000:9.4f} mm  tiles worked {worked}/{seen}  "
              f"sparse {c.get_option(OPT_SPARSE)}{wet}", flush=True)
        if md < eltol:
            break
print(f"total {time.perf_counter() - t_all:.1f} s")

#!/usr/bin/env python3
"""time the stencil kernels on non-square rasters (e.g. the slabs of a multi-GPU run):
    shape_bench.py rows cols [iters] [kernels: fused,pass] [module: add|drain]
The DEM is rounded to 1e-4 m, so the DEM-code path applies as it does to real DEMs (WDPM_DEM32=0 turns it off)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wdpm_amd
R, C = int(sys.argv[1]), int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
module = sys.argv[5] if len(sys.argv) > 5 else "add"
lib = wdpm_amd.load_hip()
rng = np.random.default_rng(1)
bd = np.full((R + 2, C + 2), -99999.0); bd[1:-1, 1:-1] = np.round(500 + rng.random((R, C)), 4)
bw = np.where(bd > -99999.0, 0.1, 0.0)
kw = {}
if module == "drain":
    k = int(np.argmin(np.where(bd > 0, bd, np.inf))); kw = dict(drainrow=k // (C + 2), draincol=k % (C + 2))
for name, k in (("fused", wdpm_amd.KERNEL_FUSED), ("pass", wdpm_amd.KERNEL_PASS)):
    if only and name not in only: continue
    with lib.context(module=module, nrows=R, ncols=C, missingvalue=-99999.0, kernel=k, **kw) as c:
        c.upload(bd, bw); c.run_block(20, 0.0); c.iterate(12); c.synchronize(); c.timing_reset()   # a block first: the library learns that the raster is wet (wdpm_max_diff), and what its XCDs deliver
        t = time.perf_counter(); c.iterate(iters); c.synchronize(); dt = time.perf_counter() - t
        n, ms = c.timing()
        print(f"{R}x{C} {module} {name:8s} dem32={c.get_option(wdpm_amd.OPT_DEM32)} {R*C*iters/dt:.4g} cell-updates/s  {ms/iters*1000:.1f} us/iteration")

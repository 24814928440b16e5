#!/usr/bin/env python3
"""How much of a small raster's drain iteration is the outlet's slow path?  The same raster and water with the outlet where
WDPMCL.c:1005-1017 puts it and with an outlet that lies outside the raster (no wave then takes the sink's path)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wdpm_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 482
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
lib = wdpm_amd.load_hip()
dem = lib.synth_dem(n, n)
bd = np.full((n + 2, n + 2), -99999.0); bd[1:-1, 1:-1] = dem
bw = np.where(bd > -99999.0, 0.1, 0.0)
k = int(np.argmin(np.where(bd > 0, bd, np.inf)))
cases = (("drain, outlet in the raster", dict(module="drain", drainrow=k // (n + 2), draincol=k % (n + 2))),
         ("drain, no outlet", dict(module="drain", drainrow=-500, draincol=-500)),
         ("add", dict(module="add")))
for name, kw in cases:
    with lib.context(nrows=n, ncols=n, missingvalue=-99999.0, **kw) as c:
        c.upload(bd, bw)
        c.run_block(50, 1e-6)
        c.iterate(50); c.synchronize()
        t = time.perf_counter(); c.iterate(iters); c.synchronize(); dt = time.perf_counter() - t
        print(f"{n}x{n} {name:30s} {dt / iters * 1e6:7.2f} us per iteration  (gate-free variants: {c.get_option(wdpm_amd.capi.OPT_PLAIN_WATER)})")

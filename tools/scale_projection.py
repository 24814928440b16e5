#!/usr/bin/env python3
"""What ONE rank of an N-GPU run does between two halo refreshes, timed on one GPU: the compute side of the scaling
curve (a projection - the halo transfers themselves need a second GPU).  For N = 1, 2, 4, 8 the slab of the middle
rank of the 16384^2 add workload (halos for k iterations on both sides) runs k-1 plain iterations and one overlapped
iteration (three launches, the interior on the side stream), exactly as wdpm_rank_iterate queues them.
    scale_projection.py [size] [k] [groups] [plain]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wdpm_amd
from wdpm_amd.rowblock import partition

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 50
plain = len(sys.argv) > 4 and sys.argv[4] == "plain"        # no overlapped iteration: the refresh would wait for the whole group
MISS = -99999.0
lib = wdpm_amd.load_hip()
dem = lib.synth_dem(n, n)
base = None
for N in (1, 2, 4, 8):
    slabs = partition(lib, n, N, k)
    s = slabs[N // 2]
    bd = np.full((s.rows, n + 2), MISS)
    lo, hi = max(s.row0, 1), min(s.row0 + s.rows, n + 1)
    bd[lo - s.row0:hi - s.row0, 1:-1] = dem[lo - 1:hi - 1]
    bw = np.where(bd > MISS, 0.1, 0.0)
    with lib.context(module="add", nrows=n, ncols=n, missingvalue=MISS, slab_row0=s.row0, slab_rows=s.rows if N > 1 else 0) as c:
        c.upload(bd, bw)
        top = s.lo + slabs[s.rank - 1].down if s.rank > 0 else 0
        bottom = s.rows - (s.hi - slabs[s.rank + 1].up) if s.rank < N - 1 else 0
        def group():
            if N > 1 and not plain:
                c.iterate_overlapped(k, top, bottom)
            else:
                c.iterate(k)
        c.run_block(2 * k, 0.0)          # a block first, as every real run has had by then: the library learns that the raster is wet
        for _ in range(6):               # (wdpm_max_diff) and what the XCDs deliver (chunk heights, round 4)
            group()
        c.synchronize()
        t = time.perf_counter()
        for _ in range(groups):
            group()
        c.synchronize()
        dt = (time.perf_counter() - t) / (groups * k)
    if base is None:
        base = dt
    halo_mb = (11 + 22) * (n + 2) * 8 * (2 if 0 < s.rank < N - 1 else 1) / 1e6 * (3 * k - 1 + 6 * k - 2) / 33
    print(f"N={N}: slab {s.rows} x {n + 2} (owns {s.own_hi - s.own_lo + 1} rows), {dt * 1e6:8.1f} us per iteration on its GPU -> "
          f"{n * n / dt:.4g} cell-updates/s aggregate if the halo refresh is hidden, x{base / dt:.2f} of one GPU; "
          f"per refresh and rank {halo_mb:.1f} MB sent + as much received every {k} iterations "
          f"(~{halo_mb / 2 / 150e3 * 1e6 + 20:.0f} us at 150 GB/s per direction + latency, i.e. {100 * (halo_mb / 2 / 150e3 * 1e6 + 20) / (k * dt * 1e6):.1f} % of a group if NOT hidden)")

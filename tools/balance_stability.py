#!/usr/bin/env python3
"""Does the adaptive chunk-height balance (wdpm_kernels.h::XcdBalance) stay put over a long run?  Blocks of the reference's loop on the
synthetic raster, ms per iteration of each block (HIP events of the stencil launches), with WDPM_BALANCE=1 (default) and 0.
    balance_stability.py [size=16384] [blocks=12] [iterations per block=250]"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, root)
    import numpy as np
    import wdpm_amd
    n, blocks, iters = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    lib = wdpm_amd.load_hip()
    dem = lib.synth_dem(n, n)
    bd = np.full((n + 2, n + 2), -99999.0); bd[1:-1, 1:-1] = dem
    bw = np.where(bd > -99999.0, 0.1, 0.0)
    out = []
    with lib.context(module="add", nrows=n, ncols=n, missingvalue=-99999.0) as c:
        c.upload(bd, bw)
        for b in range(blocks):
            c.timing_reset()
            c.run_block(iters, 0.005 / 1000)
            launches, ms = c.timing()
            out.append(ms / launches)
    print(" ".join("%.4f" % v for v in out))
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "16384"
    blocks = sys.argv[2] if len(sys.argv) > 2 else "12"
    iters = sys.argv[3] if len(sys.argv) > 3 else "250"
    for bal in ("1", "0", "1", "0"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n, blocks, iters], env=dict(os.environ, WDPM_BALANCE=bal),
                           capture_output=True, text=True, timeout=900)
        print(f"{n}x{n} WDPM_BALANCE={bal}: ms per iteration, block by block: {p.stdout.strip() or p.stderr[-300:]}", flush=True)

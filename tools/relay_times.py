#!/usr/bin/env python3
"""Where does a relay-kernel launch spend its time?  Needs a timing build (make EXTRA=-DWDPM_WAVE_TIMES, WDPM_HIP_LIB): every wave leaves
eight stamps (s_memrealtime, 100 MHz, a full wait in front of each): entry, rows loaded, row alignment 1 done, past the first barrier,
alignment 2 done, past the second barrier, alignment 3 done, stores acknowledged.     relay_times.py [size=482] [module=add]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wdpm_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 482
module = sys.argv[2] if len(sys.argv) > 2 else "add"
lib = wdpm_amd.load_hip()
raw = ctypes.CDLL(os.environ["WDPM_HIP_LIB"])
rng = np.random.default_rng(1)
bd = np.full((n + 2, n + 2), -99999.0); bd[1:-1, 1:-1] = np.round(500 + rng.random((n, n)), 4)
bw = np.where(bd > -99999.0, 0.1, 0.0)
kw = {}
if module == "drain":
    k = int(np.argmin(np.where(bd > 0, bd, np.inf))); kw = dict(drainrow=k // (n + 2), draincol=k % (n + 2))
names = ["entry", "rows loaded", "alignment 1 done", "past barrier 1", "alignment 2 done", "past barrier 2", "alignment 3 done", "stores acknowledged"]
with lib.context(module=module, nrows=n, ncols=n, missingvalue=-99999.0, kernel=wdpm_amd.KERNEL_FUSED, **kw) as c:
    c.upload(bd, bw); c.iterate(200); c.synchronize()
    for rep in range(3):
        c.iterate(50); c.synchronize()
        buf = np.zeros((4096, 8), dtype=np.uint64)
        assert raw.wdpm_debug_relay_times(buf.ctypes.data_as(ctypes.c_void_p), 8192) == 0
        t = buf[(buf[:, 0] > 0) & (buf[:, 7] >= buf[:, 0]) & (buf[:, 7] - buf[:, 0] < 10**7)].astype(np.int64)
        base = t[:, 0].min(); u = (t - base) / 100.0
        print(f"== {n}x{n} {module}, launch {rep}: {len(t)} waves, span {u[:, 7].max():.2f} us")
        for k in range(8):
            col = u[:, k]
            print("   %-22s p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (names[k], *np.percentile(col, [10, 50, 90, 100])))
        d = np.diff(u, axis=1)
        print("   per wave, median of each interval:", " ".join("%.2f" % x for x in np.median(d, axis=0)))

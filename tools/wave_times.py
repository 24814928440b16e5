#!/usr/bin/env python3
"""When do the waves of one marching-kernel launch start and end?  Needs a timing build of the library
(make EXTRA=-DWDPM_WAVE_TIMES, pointed at with WDPM_HIP_LIB): every wave leaves its start / end time
(s_memrealtime, 100 MHz), its XCD and its strip / chunk in a device array, read back here after a steady
launch.      wave_times.py [size=16384] [rows=size] [module=add]
Prints the launch's span, how the ends spread (the tail is what a one-round launch pays for imbalance),
and the same per XCD."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wdpm_amd
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
R = int(sys.argv[2]) if len(sys.argv) > 2 else C
module = sys.argv[3] if len(sys.argv) > 3 else "add"
lib = wdpm_amd.load_hip()
raw = ctypes.CDLL(os.environ["WDPM_HIP_LIB"])
rng = np.random.default_rng(1)
bd = np.full((R + 2, C + 2), -99999.0); bd[1:-1, 1:-1] = np.round(500 + rng.random((R, C)), 4)
bw = np.where(bd > -99999.0, 0.1, 0.0)
kw = {}
if module == "drain":
    k = int(np.argmin(np.where(bd > 0, bd, np.inf))); kw = dict(drainrow=k // (C + 2), draincol=k % (C + 2))
row0 = int(os.environ.get("WT_ROW0", "0"))      # > 0 (a multiple of 3): the same rows as a MIDDLE slab of a taller raster - no NODATA border row on top
if row0:
    bd[0, 1:-1] = np.round(500 + rng.random(C), 4); bw[0, 1:-1] = 0.1; bd[-1, 1:-1] = np.round(500 + rng.random(C), 4); bw[-1, 1:-1] = 0.1
    kw.update(slab_row0=row0, slab_rows=R + 2)
with lib.context(module=module, nrows=(R + 2 * row0 if row0 else R), ncols=C, missingvalue=-99999.0, kernel=wdpm_amd.KERNEL_FUSED, **kw) as c:
    c.upload(bd, bw); c.run_block(20, 0.0); c.iterate(int(os.environ.get('WT_WARM', '30'))); c.synchronize()
    c.timing_reset(); c.iterate(100); c.synchronize()
    period_us = c.timing()[1] / 100 * 1000          # launch to launch on the stream, this build, this box (HIP events around 100 iterations)
    print(f"period of a launch on the stream: {period_us:.1f} us per iteration (compare with the spans below: the rest is dispatch, ramp and end-of-kernel)")
    prev_wg = None
    for rep in range(3):
        c.iterate(5); c.synchronize()
        buf = np.zeros((8192, 4), dtype=np.uint64)
        rc = raw.wdpm_debug_wave_times(buf.ctypes.data_as(ctypes.c_void_p), 8192)
        assert rc == 0
        t = buf[buf[:, 1] > 0]
        n = len(t)
        t0 = t[:, 0].astype(np.int64); t1 = t[:, 1].astype(np.int64)
        base = t0.min(); s = (t0 - base) / 100.0; e = (t1 - base) / 100.0   # us
        span = e.max()
        xcc = (t[:, 2] >> np.uint64(32)).astype(int) & 15
        strip = (t[:, 3] >> np.uint64(48)).astype(int); chunk = (t[:, 3] & np.uint64(0xffffffff)).astype(int)
        nst = ((t[:, 3] >> np.uint64(32)) & np.uint64(0xffff)).astype(int)       # marching steps of the wave (H / 3 + 2)
        dur = e - s
        print(f"== {R}x{C} {module}, launch {rep}: {n} waves ({strip.max()+1} strips x {chunk.max()+1} chunks), span {span:.1f} us")
        print("   starts  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us" % tuple(np.percentile(s, [50, 90, 99, 100])))
        print("   ends    p1 %.1f  p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us" % tuple(np.percentile(e, [1, 10, 50, 90, 99, 100])))
        print("   durations  min %.1f  p50 %.1f  p90 %.1f  max %.1f us;  wave-time in flight / (waves x span) = %.3f" %
              (dur.min(), np.median(dur), np.percentile(dur, 90), dur.max(), dur.sum() / (n * span)))
        slot = (t[:, 2] & np.uint64(15)).astype(int); simd = ((t[:, 2] >> np.uint64(4)) & np.uint64(3)).astype(int)
        print("   wave slots used:", {int(k): int((slot == k).sum()) for k in np.unique(slot)},
              " median end by slot:", {int(k): round(float(np.median(e[slot == k])), 1) for k in np.unique(slot)})
        for x in range(8):
            m = xcc == x
            if m.any():
                print("   XCD %d: %4d waves, median duration %.1f, mean %.1f, last end %.1f us; steps per wave mean %.2f (%s); us per step %.3f" %
                      (x, m.sum(), np.median(dur[m]), dur[m].mean(), e[m].max(), nst[m].mean(), dict(zip(*np.unique(nst[m], return_counts=True))), dur[m].sum() / nst[m].sum()))
        print("   balance:", c.balance_info())
        # logical XCD of a work item (the kernel's remap: item -> workgroup vb -> XCD share vb // (grid / 8), grid a multiple of 8) against
        # the physical one it ran on: a constant difference = the dispatcher's round-robin started at another XCD in this launch
        item = np.nonzero(buf[:, 1] > 0)[0]; wpb = 8 if len(np.unique(slot)) > 1 else 4
        grid = ((int(item.max()) + wpb) // wpb + 7) // 8 * 8
        rot = (xcc - (item // wpb) // (grid // 8)) % 8
        print("   physical - logical XCD (mod 8):", dict(zip(*np.unique(rot, return_counts=True))))
        # per SIMD (XCD, the hardware id above the SIMD bits, SIMD): how many waves it held, when its last one ended - a launch of one
        # resident round ends with its busiest SIMD (round 5: tall chunks paired with short ones per SIMD)
        hw = (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
        key = xcc.astype(np.int64) * (1 << 32) + (hw >> 8) * 4 + simd
        uk, inv = np.unique(key, return_inverse=True)
        cnt = np.bincount(inv); last_end = np.zeros(len(uk)); np.maximum.at(last_end, inv, e)
        tot = np.bincount(inv, weights=dur)
        print("   SIMDs holding waves: %d (1 wave: %d, 2: %d, more: %d); a SIMD's last end  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us; sum of its waves' durations p50 %.1f max %.1f" %
              (len(uk), (cnt == 1).sum(), (cnt == 2).sum(), (cnt > 2).sum(), *np.percentile(last_end, [10, 50, 90, 100]), np.median(tot), tot.max()))
        by_chunk = [round(float(np.median(dur[chunk == k])), 1) for k in range(chunk.max() + 1)]
        print("   median duration by chunk row:", by_chunk[:12], "..." if len(by_chunk) > 24 else "", by_chunk[12:] if len(by_chunk) <= 24 else by_chunk[-12:])
        # is a slow workgroup slow again in the next launch (same table, same geometry)?  Mean duration per step of each workgroup's waves,
        # this launch against the previous one looked at, and whether the workgroup sat on the same CU (hardware id above the wave-slot bits)
        full = np.zeros(len(buf)); full[item] = dur / np.maximum(nst, 1)
        hwfull = np.zeros(len(buf), dtype=np.int64); hwfull[item] = (xcc.astype(np.int64) << 32) | (hw >> 6)
        nwg = (int(item.max()) + wpb) // wpb
        per_wg = np.array([full[w * wpb:(w + 1) * wpb][full[w * wpb:(w + 1) * wpb] > 0].mean() if (full[w * wpb:(w + 1) * wpb] > 0).any() else 0.0 for w in range(nwg)])
        cu_of = np.array([hwfull[w * wpb] for w in range(nwg)])
        if prev_wg is not None and len(prev_wg[0]) == len(per_wg):
            ok = (per_wg > 0) & (prev_wg[0] > 0)
            same_cu = float((cu_of[ok] == prev_wg[1][ok]).mean())
            r = float(np.corrcoef(per_wg[ok], prev_wg[0][ok])[0, 1])
            rel = per_wg[ok] / np.median(per_wg[ok])
            print("   workgroups: us per step  p5 %.3f  p50 %.3f  p95 %.3f  max %.3f (x median: %.3f .. %.3f); correlation with the previous launch looked at %.2f; on the same CU as then: %.0f %%" %
                  (*np.percentile(per_wg[ok], [5, 50, 95, 100]), rel.min(), rel.max(), r, 100 * same_cu))
        prev_wg = (per_wg, cu_of)
        last = np.argsort(e)[-8:]
        print("   last to end (strip, chunk, xcd, start, end):", [(int(strip[i]), int(chunk[i]), int(xcc[i]), round(float(s[i]), 1), round(float(e[i]), 1)) for i in last])
        first = np.argsort(e)[:4]
        print("   first to end:", [(int(strip[i]), int(chunk[i]), int(xcc[i]), round(float(s[i]), 1), round(float(e[i]), 1)) for i in first])

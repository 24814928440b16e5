#!/usr/bin/env python3
"""Does a HIP graph shorten the launch-to-launch period of the iteration kernel?  An experiment, not product code: the library's
ordinary launches of K iterations are captured from the context's stream (hipStreamBeginCapture on a stream handed to the library
with wdpm_set_stream), instantiated once and replayed; timed against the same K iterations launched the ordinary way.
    graph_probe.py rows cols [module=add] [K=48]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wdpm_amd
R, Cc = int(sys.argv[1]), int(sys.argv[2]); module = sys.argv[3] if len(sys.argv) > 3 else "add"; K = int(sys.argv[4]) if len(sys.argv) > 4 else 48
lib = wdpm_amd.load_hip()
hip = lib.dll            # the HIP runtime the library itself is bound to: dlsym on its handle searches its dependencies
for _n in ("hipStreamCreate", "hipStreamBeginCapture", "hipStreamEndCapture", "hipGraphInstantiate", "hipGraphLaunch", "hipGraphExecDestroy", "hipGraphDestroy"):
    getattr(hip, _n).restype = C.c_int
def ck(e, what):
    if e != 0: raise SystemExit(f"{what} failed: hip error {e}")
rng = np.random.default_rng(1)
bd = np.full((R + 2, Cc + 2), -99999.0); bd[1:-1, 1:-1] = np.round(500 + rng.random((R, Cc)), 4)
bw = np.where(bd > -99999.0, 0.1, 0.0)
kw = {}
if module == "drain":
    k = int(np.argmin(np.where(bd > 0, bd, np.inf))); kw = dict(drainrow=k // (Cc + 2), draincol=k % (Cc + 2))
with lib.context(module=module, nrows=R, ncols=Cc, missingvalue=-99999.0, kernel=wdpm_amd.KERNEL_FUSED, **kw) as c:
    s = C.c_void_p()
    ck(hip.hipStreamCreate(C.byref(s)), "hipStreamCreate")
    c.upload(bd, bw); c.set_stream(s.value); c.run_block(20, 0.0); c.iterate(400); c.synchronize()     # the XCD balance has converged
    def timed(fn, reps=12):
        best = []
        for _ in range(reps):
            c.synchronize(); t = time.perf_counter(); fn(); c.synchronize(); best.append(time.perf_counter() - t)
        best.sort(); return best[len(best) // 2] / K * 1e6
    plain = timed(lambda: c.iterate(K))
    graph, gexec = C.c_void_p(), C.c_void_p()
    ck(hip.hipStreamBeginCapture(s, 2), "hipStreamBeginCapture")          # hipStreamCaptureModeRelaxed
    c.iterate(K)
    ck(hip.hipStreamEndCapture(s, C.byref(graph)), "hipStreamEndCapture")
    ck(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")
    ck(hip.hipGraphLaunch(gexec, s), "hipGraphLaunch"); c.synchronize()
    replay = timed(lambda: ck(hip.hipGraphLaunch(gexec, s), "hipGraphLaunch"))
    plain2 = timed(lambda: c.iterate(K))
    print(f"{R}x{Cc} {module}: {K} iterations launched one by one {plain:.2f} / {plain2:.2f} us per iteration, replayed as one HIP graph {replay:.2f} us ({(plain2 / replay - 1) * 100:+.1f} %)")
    hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph)

# experiment only: are the first iterations of a run slower because of the set-up (fresh allocations, an upload just done) or because
# of the STATE (0.1 m of water on every cell, none moved yet)?  Same context, same buffers: time 5 + 20 iterations after uploading
# the initial water and after uploading a state 130 iterations old.
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, wdpm_amd
lib = wdpm_amd.load_hip()
n = int(os.environ.get("N", "16384"))
dem = lib.synth_dem(n, n)
bd = np.full((n + 2, n + 2), -99999.0); bd[1:-1, 1:-1] = dem; del dem
bw0 = np.where(bd > -99999.0, 0.1, 0.0)

def timed(c, warm, steps):
    c.run_block(warm, 5e-6)
    c.timing_reset()
    t = time.perf_counter(); c.run_block(steps, 5e-6); dt = time.perf_counter() - t
    la, ms = c.timing_steady()
    return ms / la, dt / steps * 1e3

with lib.context(module="add", nrows=n, ncols=n, missingvalue=-99999.0) as c:
    c.upload(bd, bw0); c.synchronize()
    print("A fresh context, initial water, 5 + 20:          kernel %.4f ms, ms/step %.4f" % timed(c, 5, 20), flush=True)
    print("B the same context going on (iterations 26-45):  kernel %.4f ms, ms/step %.4f" % timed(c, 0, 20), flush=True)
    print("  ... 46-65:                                     kernel %.4f ms, ms/step %.4f" % timed(c, 0, 20), flush=True)
    timed(c, 0, 65)
    old = c.download_water().copy()
    print("  water after 130 iterations: %.1f %% of the cells dry, max %.3f m" % (100.0 * np.mean(old[1:-1, 1:-1] == 0.0), old.max()), flush=True)
    c.upload_water(bw0); c.synchronize()
    print("C same context, initial water uploaded again:     kernel %.4f ms, ms/step %.4f" % timed(c, 5, 20), flush=True)
    c.upload_water(old); c.synchronize()
    print("D same context, the 130-iteration state uploaded: kernel %.4f ms, ms/step %.4f" % timed(c, 5, 20), flush=True)
    c.upload_water(bw0); c.synchronize()
    for lo in range(0, 60, 5):
        k, s = timed(c, 0, 5)
        print("E initial water again, iterations %2d-%2d: kernel %.4f ms" % (lo + 1, lo + 5, k), flush=True)
with lib.context(module="add", nrows=n, ncols=n, missingvalue=-99999.0) as c:
    c.upload(bd, old); c.synchronize()
    print("F fresh context, the 130-iteration state, 5 + 20: kernel %.4f ms, ms/step %.4f" % timed(c, 5, 20), flush=True)

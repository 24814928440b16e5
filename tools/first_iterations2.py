# experiment only: what slows the iterations that follow an upload - the idle time of the compute units, or something a burst of
# memory traffic repairs?  One context, its buffers untouched; between measurements only sleeps and device-to-device copies.
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, wdpm_amd
lib = wdpm_amd.load_hip()
n = 16384
dem = lib.synth_dem(n, n)
bd = np.full((n + 2, n + 2), -99999.0); bd[1:-1, 1:-1] = dem; del dem
bw0 = np.where(bd > -99999.0, 0.1, 0.0)
a = torch.empty(1 << 28, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)    # 2 GiB each

def blocks(c, k=6):
    out = []
    for _ in range(k):
        c.timing_reset(); c.run_block(5, 5e-6); la, ms = c.timing_steady(); out.append(ms / la)
    return " ".join("%.3f" % x for x in out)

def copies(ms):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        b.copy_(a); torch.cuda.synchronize()

def sweeps(c, ms):      # the marching kernel itself on another context's rasters
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        c.run_block(5, 5e-6)

with lib.context(module="add", nrows=n, ncols=n, missingvalue=-99999.0) as c, lib.context(module="add", nrows=4096, ncols=n, missingvalue=-99999.0) as c2:
    c.upload(bd, bw0); c.synchronize()
    c2.upload(bd[:4098].copy(), bw0[:4098].copy()); c2.synchronize()
    print("kernel ms, six blocks of 5 iterations each")
    print("after the upload:                          ", blocks(c), flush=True)
    print("straight on:                               ", blocks(c), flush=True)
    for idle in (5, 20, 100, 300):
        time.sleep(idle / 1e3)
        print("after %3d ms of nothing:                    " % idle, blocks(c), flush=True)
    for ms in (10, 30, 100):
        time.sleep(0.3); copies(ms)
        print("300 ms of nothing, %3d ms of 2 GiB copies:  " % ms, blocks(c), flush=True)
    for ms in (10, 30, 100):
        time.sleep(0.3); sweeps(c2, ms)
        print("300 ms of nothing, %3d ms of the kernel on another raster: " % ms, blocks(c), flush=True)
    c.upload_water(bw0); c.synchronize()
    print("after an upload of the water:              ", blocks(c), flush=True)
    c.upload_water(bw0); c.synchronize(); copies(30)
    print("upload, 30 ms of copies:                   ", blocks(c), flush=True)
    c.upload_water(bw0); c.synchronize(); sweeps(c2, 30)
    print("upload, 30 ms of the kernel elsewhere:     ", blocks(c), flush=True)

# experiment only: how long does the card keep its clocks after a burst of compute?  (decides whether set-up work spread over the
# upload could have the timed block of `--warmup 5 --steps 20` start on raised clocks)
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, wdpm_amd
lib = wdpm_amd.load_hip()
n = 16384
dem = lib.synth_dem(n, n)
bd = np.full((n + 2, n + 2), -99999.0); bd[1:-1, 1:-1] = dem; del dem
bw = np.where(bd > -99999.0, 0.1, 0.0)
def run(burst_ms, idle_ms):
    with lib.context(module="add", nrows=n, ncols=n, missingvalue=-99999.0) as c:
        c.upload(bd, bw)
        c.synchronize()
        if burst_ms:
            a = torch.randn(8192, 8192, device="cuda", dtype=torch.float32)
            t = time.perf_counter()
            while (time.perf_counter() - t) * 1e3 < burst_ms:
                a = (a @ a).clamp_(-1, 1); torch.cuda.synchronize()
            del a
        if idle_ms:
            time.sleep(idle_ms / 1e3)
        c.run_block(5, 5e-6)
        c.timing_reset()
        t = time.perf_counter(); c.run_block(20, 5e-6); dt = time.perf_counter() - t
        la, ms = c.timing_steady()
        return ms / la, dt / 20 * 1e3
for burst, idle in ((0, 0), (60, 0), (60, 20), (60, 100), (60, 500), (200, 0), (0, 0)):
    k, step = run(burst, idle)
    print(f"burst {burst:3d} ms, then idle {idle:3d} ms, then --warmup 5 --steps 20: kernel {k:.4f} ms, ms/step {step:.4f}", flush=True)

# experiment only: what a pure HBM stream draws on this board (rocm-smi polled beside a torch fp64 triad, 2 reads : 1 write, the mix
# of the stencil), for the two-currencies estimate of DESIGN.md 9.  usage on the GPU box: python tools/power_probe.py
import re, statistics, subprocess, threading, time
import torch
def poll():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out); p = re.search(r"Graphics Package Power \(W\): ([0-9.]+)", out)
    return (int(s.group(1)) if s else None, float(p.group(1)) if p else None)
n = 1 << 28
a = torch.rand(n, dtype=torch.float64, device="cuda"); b = torch.rand(n, dtype=torch.float64, device="cuda"); c = torch.empty_like(a)
samples, stop = [], False
def watcher():
    while not stop:
        samples.append((time.perf_counter(), *poll()))
th = threading.Thread(target=watcher); th.start()
time.sleep(1.0)
t0 = time.perf_counter(); it = 0
while time.perf_counter() - t0 < 4.0:
    for _ in range(50):
        torch.add(a, b, out=c)
    torch.cuda.synchronize(); it += 50
t1 = time.perf_counter()
time.sleep(0.5); stop = True; th.join()
print("triad: %.0f GB/s of real traffic over %.1f s" % (it * 3 * 8 * n / (t1 - t0) / 1e9, t1 - t0))
hot = [(s, p) for t, s, p in samples if s and p and t0 + 1.5 < t < t1 - 0.2]
print("during it: sclk median %s MHz, power median %s W (max %s), %d samples" % (statistics.median(s for s, _ in hot), statistics.median(p for _, p in hot), max(p for _, p in hot), len(hot)))
idle = [(s, p) for t, s, p in samples if s and p and t < t0 - 0.2]
print("before it: sclk %s MHz, power %s W" % (statistics.median(s for s, _ in idle) if idle else None, statistics.median(p for _, p in idle) if idle else None))

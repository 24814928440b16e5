# experiment only: median shader clock and board power (rocm-smi, polled) while bench.py runs long blocks at several raster sizes and
# modules - is a small launch at the board's power cap too?  (round 5: at 16384^2 an idle tail is clock for the rest; is that so at
# 4096^2?)          usage on the GPU box: python tools/clock_watch_sizes.py > gpurun_out/r05/clock_watch_sizes.txt
import json, os, re, statistics, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
CASES = [("add 16384^2", ["--size", "16384", "--steps", "3000"]), ("add 8192^2", ["--size", "8192", "--steps", "12000"]),
         ("add 4096^2", ["--size", "4096", "--steps", "40000"]), ("add 2048^2", ["--size", "2048", "--steps", "100000"]),
         ("drain 8192^2", ["--module", "drain", "--size", "8192", "--steps", "9000", "--drain-spinup", "200"]),
         ("drain 4096^2", ["--module", "drain", "--size", "4096", "--steps", "30000", "--drain-spinup", "200"])]

def poll():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    p = re.search(r"Graphics Package Power \(W\): ([0-9.]+)", out)
    return (int(s.group(1)) if s else None, float(p.group(1)) if p else None)

for name, args in CASES:
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--warmup", "20", "--no-cpu-baseline"],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
    samples = []
    while child.poll() is None:
        samples.append(poll())
    line = child.stdout.read().strip().splitlines()[-1]
    d = json.loads(line)
    hot = [(s, p) for s, p in samples if s and p and p > 600.0]
    k = len(hot) // 5
    mid = hot[k:len(hot) - k] if len(hot) >= 10 else hot
    print("%-14s kernel %.4f ms  job_frac %.3f  sclk median %s MHz (min %s max %s)  power median %s W (max %s)  [%d samples]" % (
        name, d["roofline"]["kernel_ms_per_iteration"], d["roofline"]["job_frac"],
        statistics.median(s for s, _ in mid) if mid else None, min((s for s, _ in mid), default=None), max((s for s, _ in mid), default=None),
        statistics.median(p for _, p in mid) if mid else None, max((p for _, p in mid), default=None), len(mid)), flush=True)
    time.sleep(1.0)

# experiment only: median shader clock and board power (rocm-smi, polled) while bench.py runs a long block, for several settings of
# the library's A/B switches - what a change in bytes or in waves per SIMD does to the clock the card holds at its power cap.
# usage on the GPU box: python tools/clock_watch.py > gpurun_out/r04/clock_watch_variants.txt
import json, os, re, statistics, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
VARIANTS = [("default (16-bit DEM offsets, two waves per SIMD)", {}),
            ("32-bit DEM codes (WDPM_DEM16=0)", {"WDPM_DEM16": "0"}),
            ("fp64 DEM (WDPM_DEM32=0)", {"WDPM_DEM32": "0"}),
            ("no issue priorities (WDPM_PRIO=0)", {"WDPM_PRIO": "0"}),
            ("unclamped step (WDPM_CLAMP=0)", {"WDPM_CLAMP": "0"}),
            ("default again", {})]

def poll():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    p = re.search(r"Graphics Package Power \(W\): ([0-9.]+)", out)
    return (int(s.group(1)) if s else None, float(p.group(1)) if p else None)

for name, env in VARIANTS:
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3000", "--warmup", "20", "--no-cpu-baseline"],
                             env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
    samples = []
    while child.poll() is None:
        samples.append(poll())
    line = child.stdout.read().strip().splitlines()[-1]
    d = json.loads(line)
    hot = [(s, p) for s, p in samples if s and p and p > 1000.0]
    # the read-out of the power is a moving average: drop the first and last fifth of the samples above 1000 W
    k = len(hot) // 5
    mid = hot[k:len(hot) - k] if len(hot) >= 10 else hot
    print("%-52s kernel %.4f ms  job_frac %.3f  sclk median %s MHz (min %s max %s)  power median %s W  [%d samples]" % (
        name, d["roofline"]["kernel_ms_per_iteration"], d["roofline"]["job_frac"],
        statistics.median(s for s, _ in mid) if mid else None, min((s for s, _ in mid), default=None), max((s for s, _ in mid), default=None),
        statistics.median(p for _, p in mid) if mid else None, len(mid)), flush=True)
    time.sleep(1.0)

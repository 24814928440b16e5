#!/bin/bash
# interleaved A/B of HIP library builds on one box: usage ab_interleaved.sh <rounds> <name>...  ("base" = the tree's library, otherwise
# wdpm_amd/csrc/alt_<name>_libwdpm_hip.so; NAME=VALUE words are exported for that variant, e.g. "base WDPM_CLAMP=0").  Extra bench
# arguments through BENCH_ARGS.  Prints every run's kernel time and the per-variant minimum and median.
cd $GRAFT_REPO_ROOT
rounds=$1; shift
python - "$rounds" "$@" <<'PY'
import json, os, statistics, subprocess, sys
rounds = int(sys.argv[1]); variants = sys.argv[2:]
extra = os.environ.get("BENCH_ARGS", "--steps 100 --warmup 5").split()
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        env = dict(os.environ)
        words = v.split()
        if words[0] != "base":
            env["WDPM_HIP_LIB"] = os.path.join(os.getcwd(), "wdpm_amd/csrc/alt_%s_libwdpm_hip.so" % words[0])
        for w in words[1:]:
            k, val = w.split("=", 1); env[k] = val
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=300)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            k = d["roofline"]["kernel_ms_per_iteration"]; res[v].append((k, d["ms_per_step"]))
            print("round %d  %-28s kernel %.4f ms  step %.4f ms  value %.4g" % (r, v, k, d["ms_per_step"], d["value"]), flush=True)
        except Exception as e:
            print("round %d  %-28s FAILED %s %s" % (r, v, e, out.stderr[-300:]), flush=True)
for v in variants:
    ks = [a for a, _ in res[v]]
    if ks: print("== %-28s kernel min %.4f  median %.4f ms   step median %.4f ms" % (v, min(ks), statistics.median(ks), statistics.median([b for _, b in res[v]])))
PY

#!/usr/bin/env python3
"""interleaved A/B of HIP library builds over the shapes that matter (tools/shape_bench.py per run, one process each):
    ab_shapes.py <rounds> <variant> [<variant> ...]      variant = "base" (the tree's library) or the <name> of
    wdpm_amd/csrc/alt_<name>_libwdpm_hip.so, optionally followed by NAME=VALUE words exported for it ("base WDPM_CLAMP=0")
    SHAPES="4096x4096:add 2116x16384:add ..." overrides the default list."""
import os, re, statistics, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1]); variants = sys.argv[2:]
default = "482x471:add 482x471:drain 1000x1000:add 2000x2000:add 4096x4096:add 2116x16384:add 8192x8192:add 4096x4096:drain 1053x8190:drain 8192x8192:drain"
shapes = os.environ.get("SHAPES", default).split()
for sh in shapes:
    dims, module = sh.split(":"); R, C = dims.split("x")
    iters = "400" if int(R) * int(C) < 3000000 else "100"
    res = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            env = dict(os.environ); words = v.split()
            if words[0] != "base": env["WDPM_HIP_LIB"] = os.path.join(root, "wdpm_amd/csrc/alt_%s_libwdpm_hip.so" % words[0])
            for w in words[1:]:
                k, val = w.split("=", 1); env[k] = val
            out = subprocess.run([sys.executable, os.path.join(root, "tools/shape_bench.py"), R, C, iters, "fused", module], env=env,
                                 capture_output=True, text=True, timeout=300)
            m = re.search(r"([\d.]+) us/iteration", out.stdout)
            if m: res[v].append(float(m.group(1)))
            else: print("FAILED", sh, v, out.stderr[-300:], flush=True)
    line = "%-18s" % sh
    base = None
    for v in variants:
        if not res[v]: continue
        med = statistics.median(res[v])
        if base is None: base = med
        line += "  %s: %.2f us (min %.2f, %+.1f %%)" % (v, med, min(res[v]), (base / med - 1) * 100)
    print(line, flush=True)

#!/bin/bash
# Round-3 evidence session on one MI355X box, on the final code: rocprofv3 trace + PMC passes of the drain and small-raster
# commands (the add command's are from session E: that kernel did not change afterwards), bench lines, slab shapes, end-to-end CLI.
# Everything lands under gpurun_out/r03/ ; the summaries judged are copied to profiles/r03/ afterwards.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
bash tools/profile.sh r03/drain8192 --module drain --size 8192 --drain-spinup 2 > $O/drain8192_profile.txt 2>&1; grep -h '"metric"' $O/drain8192/trace.log | cut -c1-160
bash tools/profile.sh r03/add482 --size 482 > $O/add482_profile.txt 2>&1; grep -h '"metric"' $O/add482/trace.log | cut -c1-160
echo "== bench default"; timeout -k 10 300 python bench.py > $O/bench_default.json 2>/dev/null; cut -c1-200 $O/bench_default.json
echo "== bench --steps 20 --warmup 5"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_steps20.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'job_frac', r['job_frac'], 'hbm_real', r['hbm_real_frac'], 'valu', r['valu_issue_frac'], r['bound'])"
echo "== drain 8192"; timeout -k 10 300 python bench.py --module drain --size 8192 --steps 1000 --warmup 5 --drain-spinup 200 --no-cpu-baseline > $O/config5_drain_8192_1gpu.json 2>/dev/null; cut -c1-160 $O/config5_drain_8192_1gpu.json
echo "== config 3: 4096"; timeout -k 10 300 python bench.py --size 4096 --steps 1000 --warmup 20 --no-cpu-baseline > $O/config3_4096.json 2>/dev/null; cut -c1-160 $O/config3_4096.json
echo "== slabs of the 8-GPU runs, each alone on the GPU" | tee $O/slabs.txt
timeout -k 10 200 python tools/shape_bench.py 2049 16384 300 fused add 2>/dev/null | tee -a $O/slabs.txt
timeout -k 10 200 python tools/shape_bench.py 1055 8190 500 fused drain 2>/dev/null | tee -a $O/slabs.txt
echo "== small rasters" | tee $O/small.txt
for sz in 482 700 1000 1600; do for m in add drain; do echo -n "$m $sz: "; timeout -k 10 300 python bench.py --module $m --size $sz --steps 3000 --warmup 50 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g cell-updates/s  %.2f us per iteration' % (d['value'], d['ms_per_step']*1e3))"; done; done | tee -a $O/small.txt
echo "== end to end through WDPMCL, 16384^2"; bash tools/e2e_16k.sh > $O/e2e_cli_16384.txt 2>&1; tail -n 12 $O/e2e_cli_16384.txt
echo "== CLI basin5"; timeout -k 10 600 python -m pytest tests/test_cli.py -m gpu -q -s -k "convergence" 2>&1 | grep -E "wall|passed|failed" | tee $O/cli_basin5.txt

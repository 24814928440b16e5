#!/bin/bash
# round-2 profile session: default bench (trace + PMC), drain 8192 (trace + PMC), basin5-sized add/drain, the driver's bench commands
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02
bash tools/profile.sh r02/bench16k > gpurun_out/r02/bench16k_profile.txt 2>&1; grep -h '"metric"' gpurun_out/r02/bench16k/trace.log | cut -c1-200
bash tools/profile.sh r02/drain8192 --module drain --size 8192 --drain-spinup 2 > gpurun_out/r02/drain8192_profile.txt 2>&1; grep -h '"metric"' gpurun_out/r02/drain8192/trace.log | cut -c1-200
bash tools/profile.sh r02/add482 --size 482 > gpurun_out/r02/add482_profile.txt 2>&1; grep -h '"metric"' gpurun_out/r02/add482/trace.log | cut -c1-200
echo "== bench default"; timeout -k 10 300 python bench.py > gpurun_out/r02/bench_default.json 2>/dev/null; cut -c1-400 gpurun_out/r02/bench_default.json
echo "== bench --steps 20 --warmup 5"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_steps20.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02/bench_steps20.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['job_frac'])"
echo "== drain 8192 bench"; timeout -k 10 300 python bench.py --module drain --size 8192 --steps 1000 --warmup 5 --drain-spinup 200 --no-cpu-baseline > gpurun_out/r02/config5_drain_8192_1gpu.json 2>/dev/null; cut -c1-300 gpurun_out/r02/config5_drain_8192_1gpu.json
echo "== basin5-sized"; for m in add drain; do timeout -k 10 300 python bench.py --module $m --size 482 --steps 3000 --warmup 50 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m 482', d['value'], 'us/iter', d['ms_per_step']*1e3)"; done | tee gpurun_out/r02/size482.txt
echo "== CLI: basin5 add 300 mm to convergence + validation chain"; timeout -k 10 600 python -m pytest tests/test_cli.py -m gpu -q -s -k "convergence or validation" 2>&1 | grep -E "wall|passed|failed" | tee gpurun_out/r02/cli_basin5.txt

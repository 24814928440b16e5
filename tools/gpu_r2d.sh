#!/bin/bash
# round-2 session D: drain kernel after the straight-line fix
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r2d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_golden.py -m gpu -x -q -k "drain or golden or random" > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED" $O/pytest.log | head -20; exit 1; }
for sz in 8192 4096 1024; do echo "== drain $sz"; timeout -k 10 300 python bench.py --module drain --size $sz --steps 400 --warmup 5 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_iteration'], d['roofline']['frac'])"; done
echo "== drain slab 1055x8190"; python tools/shape_bench.py 1055 8190 300 fused drain 2>/dev/null | tail -n 3
bash tools/profile.sh r2d/drain8192 --module drain --size 8192 --drain-spinup 2 > $O/drain8192_profile.txt 2>&1; grep -E "fused_iteration_kernel<2|drain_outlet" $O/drain8192/trace/*/*_kernel_stats.csv | cut -c1-250
python - <<'P'
import json; d=json.load(open('gpurun_out/r2d/drain8192/pmc_summary.json'))
for k,v in d.items():
    if '<2' in k: print(k, {c:x['mean'] for c,x in v.items()})
P

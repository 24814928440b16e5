#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest.log | head -30; exit 1; }
for sz in 8192 1024 482; do echo "== drain $sz"; timeout -k 10 300 python bench.py --module drain --size $sz --steps 1000 --warmup 5 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_iteration'], d['roofline']['frac'])"; done
echo "== drain slab 1055x8190"; python tools/shape_bench.py 1055 8190 300 fused drain 2>/dev/null | tail -n 3

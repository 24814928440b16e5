#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r2p; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest.log | head -30; exit 1; }
for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 1000 --warmup 20"; do timeout -k 10 300 python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['steps'], d['value'], d['ms_per_step'], r['kernel_ms_per_iteration'], 'frac', r['frac'], 'job_frac', r['job_frac'])"; done

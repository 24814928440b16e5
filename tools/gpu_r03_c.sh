#!/bin/bash
# round 3, session C: the max-diff variant with its snapshot rows requested a step ahead - parity, then its duration
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_full_size_golden.py -m gpu -x -q -k "not standin" > $O/pytest_md.log 2>&1; rc=$?; tail -n 3 $O/pytest_md.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_md.log | head -40; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/md_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/md_trace.log 2>&1
cd $GRAFT_REPO_ROOT
cat $O/md_trace/*/*_kernel_stats.csv | cut -c1-60,200-400 | head -8
echo "== bench --steps 20 --warmup 5 (x3)"; for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'plain', r['kernel_ms_per_iteration'], 'all', r['kernel_ms_per_iteration_all_launches'], 'job_frac', r['job_frac'])"; done

#!/bin/bash
# round 3, session F: the DEM codes as 16-bit offsets - parity, then A/B against the 32-bit codes on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_full_size_golden.py -m gpu -x -q -k "not standin" > $O/pytest_dem16.log 2>&1; rc=$?; tail -n 3 $O/pytest_dem16.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_dem16.log | head -40; exit 1; }
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  %.4g cell-updates/s  %.2f us/iteration  kernel %.4f ms  %s' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_iteration'], r['dem'][:14]))"; }
for rep in 1 2; do for v in 0 1; do
  for n in 4096 8192; do steps=$(( 400000000 / (n * n / 1000 + 1000) )); echo -n "WDPM_DEM16=$v add $n: "; WDPM_DEM16=$v timeout -k 10 200 python bench.py --size $n --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | line; done
  echo -n "WDPM_DEM16=$v add 16384: "; WDPM_DEM16=$v timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "WDPM_DEM16=$v add 16384 steps 20 warmup 5: "; WDPM_DEM16=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line
done; done 2>&1 | tee $O/dem16_ab.txt

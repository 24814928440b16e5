#!/bin/bash
# raster-size sweep of the add kernel with the shipped defaults, DEM codes on and off (profiles/r01/size_sweep.txt)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 512 1024 2048 3072 4096 6144 8192 12288 16384; do
  steps=$(( 400000000 / (n * n / 1000 + 1000) )); [ $steps -gt 3000 ] && steps=3000; [ $steps -lt 100 ] && steps=100
  for d in 1 0; do
  WDPM_DEM32=$d timeout -k 10 200 python bench.py --size $n --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('n=%5d dem32=$d steps=%4d  %.4g cell-updates/s  %.2f us/iteration (kernel %.2f)' % ($n, d['steps'], d['value'], d['ms_per_step']*1000, d['roofline']['kernel_ms_per_iteration']*1000))"
  done
done

#!/bin/bash
# mid-size rasters: share of the resident wave slots filled (50 % = one wave per SIMD, 100 % = two) x DEM as codes or fp64
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  %.4g cell-updates/s  %.2f us/iteration' % (d['value'], d['ms_per_step']*1e3))"; }
for n in ${1:-2400 2800 3072 3600 4096 5000}; do
  steps=$(( 200000000 / (n * n / 1000 + 1000) )); [ $steps -gt 3000 ] && steps=3000
  for cfg in "X=default" "WDPM_DEM32=0 WDPM_FILL_PERCENT=50" "WDPM_DEM32=0 WDPM_FILL_PERCENT=100" "WDPM_DEM32=2 WDPM_FILL_PERCENT=50" "WDPM_DEM32=2 WDPM_FILL_PERCENT=100" "WDPM_DEM32=0 WDPM_FILL_PERCENT=75"; do
    echo -n "add n=$n $cfg: "; env $cfg timeout -k 10 200 python bench.py --size $n --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | line; done
done

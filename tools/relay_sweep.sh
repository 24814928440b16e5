#!/bin/bash
# the relay kernel forced (WDPM_RELAY=2) against the triangle / marching kernels (WDPM_RELAY=0) over raster sizes
cd $GRAFT_REPO_ROOT
for sz in ${1:-600 700 800 1000 1200 1600 2000}; do for relay in 2 0; do
  steps=$(( 300000000 / (sz * sz / 100 + 10000) )); [ $steps -gt 3000 ] && steps=3000
  echo -n "add $sz WDPM_RELAY=$relay: "; WDPM_RELAY=$relay timeout -k 10 200 python bench.py --size $sz --steps $steps --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us per iteration' % (d['ms_per_step']*1e3))"; done; done

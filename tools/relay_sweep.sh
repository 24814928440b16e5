#!/bin/bash
# the relay kernel: default choice (1), forced (2), off (0 = triangle / marching kernels) over raster sizes, add and drain
cd $GRAFT_REPO_ROOT
for m in add drain; do for sz in ${1:-482 700 1000 1200 1600}; do for relay in 1 2 0; do
  steps=$(( 300000000 / (sz * sz / 100 + 10000) )); [ $steps -gt 3000 ] && steps=3000
  echo -n "$m $sz WDPM_RELAY=$relay: "; WDPM_RELAY=$relay timeout -k 10 200 python bench.py --module $m --size $sz --steps $steps --warmup 50 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us per iteration' % (d['ms_per_step']*1e3))"; done; done; done

#!/bin/bash
# the relay kernel forced, four against eight waves per workgroup, over raster sizes (and WDPM_RELAY=0 for reference)
# usage: bash tools/relay_nw_sweep.sh "482 700 ... 8192"   (profiles/r03/relay_nw_sweep.txt: up to 2400 with both heights, 3000-8192 with eight waves)
cd $GRAFT_REPO_ROOT
for m in add drain; do for sz in ${1:-482 700 1000 1200 1600 2000 2400}; do for cfg in "WDPM_RELAY=2 WDPM_RELAY_NW=4" "WDPM_RELAY=2 WDPM_RELAY_NW=8" "WDPM_RELAY=0 WDPM_RELAY_NW=0"; do
  steps=$(( 300000000 / (sz * sz / 100 + 10000) )); [ $steps -gt 3000 ] && steps=3000
  echo -n "$m $sz $cfg: "; env $cfg timeout -k 10 200 python bench.py --module $m --size $sz --steps $steps --warmup 50 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us per iteration' % (d['ms_per_step']*1e3))"; done; done; done

#!/bin/bash
# The triangle kernel forced on rasters too big to hold all its 3-row chunks at once (several rounds of waves) against the marching kernel
# usage: bash tools/tri_sweep.sh "<sizes>" "<WDPM_TRI values>"
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  %.4g cell-updates/s  %.2f us/iteration' % (d['value'], d['ms_per_step']*1e3))"; }
for m in add drain; do for n in ${1:-1024 1280 1536 2048 2560 3072 4096}; do
  steps=$(( 200000000 / (n * n / 1000 + 1000) )); [ $steps -gt 3000 ] && steps=3000
  for t in ${2:-1 2}; do echo -n "$m n=$n WDPM_TRI=$t: "; WDPM_TRI=$t timeout -k 10 200 python bench.py --module $m --size $n --steps $steps --warmup 20 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | line; done
done; done

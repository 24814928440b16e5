#!/bin/bash
# compare kernel selections of the same library on one box
cd $GRAFT_REPO_ROOT
for k in "$@"; do
  echo "== kernel $k"
  for i in 1 2; do timeout -k 10 120 python bench.py --kernel $k --steps 100 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value %.4g  ms/step %.4f  kernel_ms/iter %.4f launches %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_iteration'], d['roofline']['launches']))"; done
done

#!/bin/bash
# round 3: the relay kernel's workgroups with an issue priority that falls from stage to stage (launches of a few rounds):
# parity with it forced everywhere, then A/B by environment: WDPM_RELAY_PRIO=0 never, 1 default (by rounds), 2 always
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
WDPM_RELAY_PRIO=2 timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_rowblock.py -m gpu -x -q > $O/pytest_relayprio.log 2>&1 || { tail -n 30 $O/pytest_relayprio.log; exit 1; }
echo "parity, forced: $(tail -n 1 $O/pytest_relayprio.log)"
us() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; }
for rep in 1 2; do for v in 0 1 2; do
  export WDPM_RELAY_PRIO=$v
  for m in add drain; do echo -n "WDPM_RELAY_PRIO=$v $m us/iteration at 482 700 850 1000 1200 1400 1600 1800 2000 2400 3000: "
    for n in 482 700 850 1000 1200 1400 1600 1800 2000 2400 3000; do timeout -k 10 200 python bench.py --module $m --size $n --steps 3000 --warmup 50 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | us; done; echo; done
done; done 2>&1 | tee $O/relay_prio_ab.txt

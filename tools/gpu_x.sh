#!/bin/bash
cd $GRAFT_REPO_ROOT
for nt in 0 1; do for m in add drain; do echo -n "WDPM_TRI_NT=$nt $m 482: "; WDPM_TRI_NT=$nt timeout -k 10 300 python bench.py --module $m --size 482 --steps 5000 --warmup 50 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us per iteration' % (d['ms_per_step']*1e3))"; done; done
for nt in 0 1; do echo -n "WDPM_TRI_NT=$nt add 1000: "; WDPM_TRI_NT=$nt timeout -k 10 300 python bench.py --size 1000 --steps 5000 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us per iteration' % (d['ms_per_step']*1e3))"; done
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "random or golden or outlet" 2>&1 | tail -n 2

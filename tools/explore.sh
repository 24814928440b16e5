#!/bin/bash
# one-off timing experiments (see profiles/r01/README.md)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout -k 10 200 python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value %.4g  kernel_ms %.4f' % (d['value'], d['roofline']['kernel_ms_per_iteration']))"; }
for fill in 50 100; do echo "drain 8192 fill $fill"; WDPM_FILL_PERCENT=$fill run --module drain --size 8192 --steps 100 --warmup 5 --drain-spinup 200; done
for fill in 50 100; do echo "drain 16384 fill $fill"; WDPM_FILL_PERCENT=$fill run --module drain --size 16384 --steps 50 --warmup 5 --drain-spinup 100; done

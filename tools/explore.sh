#!/bin/bash
# one-off timing experiments around the fused kernel's HBM behaviour (see profiles/r01/README.md)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout -k 10 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value %.4g  kernel_ms %.4f' % (d['value'], d['roofline']['kernel_ms_per_iteration']))"; }
for skew in 0 512 2048 8192 32768 131072 524288 1048576 0; do echo "skew $skew"; WDPM_ALLOC_SKEW=$skew run; done
for fill in 50 56 62 75; do echo "fill $fill"; WDPM_FILL_PERCENT=$fill run; done

#!/bin/bash
# one-off timing experiments (see profiles/r01/README.md)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value %.4g  kernel_ms %.4f' % (d['value'], d['roofline']['kernel_ms_per_iteration']))"; }
for fill in 50 100; do
  echo "base fill $fill"; WDPM_FILL_PERCENT=$fill run
  echo "dem32 fill $fill"; WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_dem32_libwdpm_hip.so WDPM_FILL_PERCENT=$fill run
done

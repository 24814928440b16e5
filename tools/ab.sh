#!/bin/bash
# A/B several builds of the HIP library: quick parity subset + bench
cd $GRAFT_REPO_ROOT
for v in "" GATE_DEM FSEL BOTH; do
  if [ -z "$v" ]; then unset WDPM_HIP_LIB; name=base; else export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so; name=$v; fi
  echo "== $name" 
  timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "golden or random or block_loop or basin5" 2>&1 | tail -n 2
  for i in 1 2; do timeout -k 10 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value %.4g  ms/step %.4f  kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_iteration']))"; done
done

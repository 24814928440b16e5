#!/bin/bash
# A/B several builds of the HIP library (timing experiments): bench only
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  if [ -z "$v" ]; then unset WDPM_HIP_LIB; name=base; else export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so; name=$v; fi
  echo "== $name"
  for i in 1 2; do timeout -k 10 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value %.4g  ms/step %.4f  kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_iteration']))"; done
done

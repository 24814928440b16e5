#!/bin/bash
# A/B alt builds for both fused kernels
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  if [ -z "$v" ]; then unset WDPM_HIP_LIB; name=base; else export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so; name=$v; fi
  for k in fused fused2; do
    echo -n "== $name $k: "
    timeout -k 10 120 python bench.py --kernel $k --steps 100 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.4g  kernel_ms/iter %.4f' % (d['value'], d['roofline']['kernel_ms_per_iteration']))"
  done
done

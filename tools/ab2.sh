#!/bin/bash
# A/B builds of the HIP library on one box: parity tests on each alternative build, then bench lines (add 8192^2 default,
# drain 8192^2, add 16384^2, add/drain 482^2).   usage: bash tools/ab2.sh <variant> ...   (wdpm_amd/csrc/alt_<variant>_libwdpm_hip.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  value %.4g  ms/step %.4f  kernel_ms %.4f  frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_per_iteration'], r['frac']))"; }
for v in "$@"; do
  export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so
  echo "== parity, $v"
  timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_golden.py tests/test_dry_tiles.py -m gpu -x -q > gpurun_out/ab/pytest_$v.log 2>&1 || { tail -n 30 gpurun_out/ab/pytest_$v.log; exit 1; }
  tail -n 1 gpurun_out/ab/pytest_$v.log
done
for rep in 1 2; do
for v in "" "$@"; do
  if [ -z "$v" ]; then unset WDPM_HIP_LIB; name=base; else export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so; name=$v; fi
  echo "== $name (pass $rep)"
  echo -n " add 8192  "; timeout -k 10 120 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | line
  echo -n " drain 8192"; timeout -k 10 120 python bench.py --module drain --size 8192 --steps 300 --warmup 5 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | line
  echo -n " add 16384 "; timeout -k 10 120 python bench.py --size 16384 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | line
  echo -n " add 482   "; timeout -k 10 120 python bench.py --size 482 --steps 3000 --warmup 50 --no-cpu-baseline 2>/dev/null | line
  echo -n " drain 482 "; timeout -k 10 120 python bench.py --module drain --size 482 --steps 3000 --warmup 50 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | line
done; done

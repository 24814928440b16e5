#!/bin/bash
# round 3, session G: A/B of alternative builds of the HIP library against the shipped one: parity on each, then bench lines
# usage: bash tools/gpu_r03_g.sh <variant> ...      (wdpm_amd/csrc/alt_<variant>_libwdpm_hip.so)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  %.4g cell-updates/s  %.2f us/iteration  kernel %.4f ms' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_iteration']))"; }
for v in "$@"; do
  export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so
  timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_full_size_golden.py -m gpu -x -q -k "not standin" > $O/pytest_$v.log 2>&1 || { tail -n 30 $O/pytest_$v.log; exit 1; }
  echo "parity $v: $(tail -n 1 $O/pytest_$v.log)"
done
for rep in 1 2; do for v in base "$@"; do
  export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so
  echo -n "$v add 482: "; timeout -k 10 200 python bench.py --size 482 --steps 3000 --warmup 50 --no-cpu-baseline 2>/dev/null | line
  for n in 2048 4096 8192; do steps=$(( 400000000 / (n * n / 1000 + 1000) )); [ $steps -gt 2000 ] && steps=2000
    echo -n "$v add $n: "; timeout -k 10 200 python bench.py --size $n --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | line; done
  echo -n "$v add 16384: "; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v drain 482: "; timeout -k 10 200 python bench.py --module drain --size 482 --steps 3000 --warmup 50 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v drain 4096: "; timeout -k 10 200 python bench.py --module drain --size 4096 --steps 500 --warmup 5 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v drain 8192: "; timeout -k 10 200 python bench.py --module drain --size 8192 --steps 300 --warmup 5 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | line
done; done 2>&1 | tee $O/ab_$1.txt

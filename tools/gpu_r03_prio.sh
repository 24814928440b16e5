#!/bin/bash
# round 3: a wave's issue priority falling with its progress (WDPM_PRIO builds): wave timing, then A/B of the variants given
# usage: bash tools/gpu_r03_prio.sh <variant> ...     (alt_<variant>_libwdpm_hip.so and alt_<variant>wt_libwdpm_hip.so)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  %.4g cell-updates/s  %.2f us/iteration  kernel %.4f ms' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_iteration']))"; }
export WDPM_LDS_PAD=36864
for v in "$@"; do
  ( WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}wt_libwdpm_hip.so timeout -k 10 200 python tools/wave_times.py 16384 &&
    WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}wt_libwdpm_hip.so timeout -k 10 200 python tools/wave_times.py 8192 8192 drain ) > $O/wave_times_$v.txt 2>&1
  echo "== $v"; grep -A4 "launch 2" $O/wave_times_$v.txt
done
for rep in 1 2; do for v in base "$@"; do
  export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so
  echo -n "$v add 8192: "; timeout -k 10 200 python bench.py --size 8192 --steps 1000 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v add 16384: "; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v slab: "; timeout -k 10 200 python tools/shape_bench.py 2049 16384 300 fused add 2>/dev/null
  echo -n "$v drain 8192: "; timeout -k 10 200 python bench.py --module drain --size 8192 --steps 300 --warmup 5 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | line
done; done 2>&1 | tee $O/prio_ab_$1.txt

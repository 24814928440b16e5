#!/bin/bash
# round 5, session 6: shares and measurements of the XCD balance by PHYSICAL XCD (BalanceArgs::rot): parity first, then against
# round 4's blockIdx % 8 (WDPM_ROT=0), interleaved, on the shapes and at 16384^2; per-XCD durations in the converged state
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s6; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_hip_parity.py tests/test_full_size_golden.py tests/test_settled_golden.py tests/test_dry_tiles.py tests/test_clamped_step.py -m gpu -q -x > $O/pytest_subset.log 2>&1 || { tail -n 30 $O/pytest_subset.log; exit 1; }
echo "parity subset: $(tail -n 1 $O/pytest_subset.log)"
SHAPES="1053x8190:drain 2049x16384:add 4096x4096:add 4096x4096:drain 8192x8192:drain 8192x8192:add" timeout -k 10 420 python tools/ab_shapes.py 3 "base" "base WDPM_ROT=0" > $O/rot_shapes_ab.txt 2>&1 || { tail $O/rot_shapes_ab.txt; exit 1; }
cat $O/rot_shapes_ab.txt
BENCH_ARGS="--steps 600 --warmup 20" timeout -k 10 300 bash tools/ab_interleaved.sh 2 base "base WDPM_ROT=0" > $O/rot_bench_ab.txt 2>&1; tail -n 2 $O/rot_bench_ab.txt
for sh in "4096 4096 add" "8190 1053 drain" "8192 8192 drain"; do
  for r in 1 0; do echo "#### WDPM_ROT=$r"; WT_WARM=300 WDPM_ROT=$r WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so timeout -k 10 120 python tools/wave_times.py $sh 2>&1 | grep -v amdgpu.ids; done
done > $O/wave_times_rot.txt 2>&1
grep -E "^####|^==|SIMDs holding|physical" $O/wave_times_rot.txt | cut -c1-200

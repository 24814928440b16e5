#!/bin/bash
# round 5, session 8: the working tree (physical-XCD shares, a class for a strip's first chunk) against HEAD (slot filling + pairing)
# and against round 4's final library, interleaved on one box: the shapes, the driver's bench; then where the waves end now
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s8; mkdir -p $O
SHAPES="1053x8190:drain 2049x16384:add 4096x4096:add 4096x4096:drain 8192x8192:drain 8192x8192:add" timeout -k 10 560 python tools/ab_shapes.py 3 base head r4 > $O/shapes_ab.txt 2>&1 || { tail $O/shapes_ab.txt; exit 1; }
cat $O/shapes_ab.txt
BENCH_ARGS="--steps 600 --warmup 20" timeout -k 10 300 bash tools/ab_interleaved.sh 2 base head r4 > $O/bench_ab.txt 2>&1; tail -n 3 $O/bench_ab.txt
for sh in "8190 1053 drain" "8192 8192 drain" "4096 4096 add"; do WT_WARM=300 WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so timeout -k 10 120 python tools/wave_times.py $sh 2>&1 | grep -v amdgpu.ids; done > $O/wave_times_first.txt 2>&1
grep -E "^==|chunk row|last to end|SIMDs" $O/wave_times_first.txt | cut -c1-300

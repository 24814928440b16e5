#!/bin/bash
# round 5, session 9: the last strip on the interior code (alt_coledge: -DWDPM_COL_EDGE_FREE) - parity first, then interleaved against the
# working tree and round 4's final library on the shapes and the bench, then where its waves end
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s9; mkdir -p $O
WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_coledge_libwdpm_hip.so timeout -k 10 500 python -m pytest tests/test_hip_parity.py tests/test_full_size_golden.py tests/test_settled_golden.py tests/test_dry_tiles.py tests/test_clamped_step.py tests/test_rowblock.py -m gpu -q > $O/pytest_coledge.log 2>&1; echo "parity subset on alt_coledge: $(tail -n 1 $O/pytest_coledge.log)"; grep -E "^FAILED|^ERROR" $O/pytest_coledge.log | head -20
SHAPES="1053x8190:drain 2049x16384:add 4096x4096:add 4096x4096:drain 8192x8192:drain 8192x8192:add" timeout -k 10 560 python tools/ab_shapes.py 3 base coledge r4 > $O/shapes_ab.txt 2>&1 || { tail $O/shapes_ab.txt; exit 1; }
cat $O/shapes_ab.txt
BENCH_ARGS="--steps 600 --warmup 20" timeout -k 10 300 bash tools/ab_interleaved.sh 2 base coledge r4 > $O/bench_ab.txt 2>&1; tail -n 3 $O/bench_ab.txt
for sh in "8190 1053 drain" "8192 8192 drain" "4096 4096 add"; do WT_WARM=300 WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wtcol_libwdpm_hip.so timeout -k 10 120 python tools/wave_times.py $sh 2>&1 | grep -v amdgpu.ids; done > $O/wave_times_coledge.txt 2>&1
grep -E "^==|last to end|SIMDs" $O/wave_times_coledge.txt | cut -c1-300

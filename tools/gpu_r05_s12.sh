#!/bin/bash
# round 5, session 12: a balance class per workgroup of work items against the ten classes (WDPM_BALANCE_WG=0): parity subset, shapes, bench, wave times
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s12; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_hip_parity.py tests/test_full_size_golden.py tests/test_settled_golden.py tests/test_dry_tiles.py tests/test_rowblock.py -m gpu -q > $O/pytest_subset.log 2>&1; echo "parity subset: $(tail -n 1 $O/pytest_subset.log)"; grep -E "^FAILED|^ERROR" $O/pytest_subset.log | head
SHAPES="1053x8190:drain 2049x16384:add 4096x4096:add 4096x4096:drain 8192x8192:drain 8192x8192:add" timeout -k 10 420 python tools/ab_shapes.py 3 "base" "base WDPM_BALANCE_WG=0" > $O/wg_shapes_ab.txt 2>&1 || { tail $O/wg_shapes_ab.txt; exit 1; }
cat $O/wg_shapes_ab.txt
BENCH_ARGS="--steps 1000 --warmup 20" timeout -k 10 300 bash tools/ab_interleaved.sh 2 base "base WDPM_BALANCE_WG=0" > $O/wg_bench_ab.txt 2>&1; tail -n 2 $O/wg_bench_ab.txt
for sh in "16384 16384 add" "8192 8192 add" "4096 4096 add"; do WT_WARM=400 WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so timeout -k 10 150 python tools/wave_times.py $sh 2>&1 | grep -v amdgpu.ids; done > $O/wave_times_wg.txt 2>&1
grep -E "^==|workgroups:|SIMDs|in flight" $O/wave_times_wg.txt | cut -c1-260

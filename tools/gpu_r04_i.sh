#!/bin/bash
# round 4: edge waves with one-offset loads, relay thresholds - the whole -m gpu suite, the kernel suites with the table forced, A/B, wave times
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_i.log 2>&1 || { tail -n 30 $O/pytest_i.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_i.log)"
WDPM_BALANCE=2 timeout -k 10 900 python -m pytest tests/test_clamped_step.py tests/test_hip_parity.py tests/test_full_size_golden.py -m gpu -x -q > $O/pytest_i2.log 2>&1 || { tail -n 30 $O/pytest_i2.log; exit 1; }
echo "suite, WDPM_BALANCE=2: $(tail -n 1 $O/pytest_i2.log)"
SHAPES="16384x16384:add 4096x4096:add 2116x16384:add 8192x8192:add 8192x8192:drain 1053x8190:drain 4096x4096:drain 3000x3000:add 482x471:add 482x471:drain" timeout -k 10 1000 python tools/ab_shapes.py 3 r3 base > $O/edge_shapes_ab.txt 2>&1; cat $O/edge_shapes_ab.txt
export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so
{ timeout -k 10 200 python tools/wave_times.py 16384; timeout -k 10 100 python tools/wave_times.py 16384 2116; timeout -k 10 100 python tools/wave_times.py 4096; } > $O/wave_times_edge2.txt 2>&1
grep -E "^==|in flight|last to end" $O/wave_times_edge2.txt

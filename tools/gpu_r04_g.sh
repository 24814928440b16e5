#!/bin/bash
# round 4: chunk heights by what each XCD delivers - kernel suites (default, and the table forced from skewed weights), A/B, per-wave timestamps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_clamped_step.py tests/test_hip_parity.py tests/test_full_size_golden.py tests/test_dry_tiles.py -m gpu -x -q > $O/pytest_g.log 2>&1 || { tail -n 30 $O/pytest_g.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_g.log)"
WDPM_BALANCE=2 timeout -k 10 900 python -m pytest tests/test_clamped_step.py tests/test_hip_parity.py tests/test_full_size_golden.py -m gpu -x -q > $O/pytest_g2.log 2>&1 || { tail -n 30 $O/pytest_g2.log; exit 1; }
echo "suite, WDPM_BALANCE=2: $(tail -n 1 $O/pytest_g2.log)"
SHAPES="16384x16384:add 4096x4096:add 2116x16384:add 8192x8192:add 8192x8192:drain 1053x8190:drain 4096x4096:drain 6000x6000:add" timeout -k 10 1000 python tools/ab_shapes.py 3 r3 "base WDPM_BALANCE=0" base > $O/balance_shapes_ab.txt 2>&1; cat $O/balance_shapes_ab.txt
export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so
{ timeout -k 10 200 python tools/wave_times.py 16384; timeout -k 10 100 python tools/wave_times.py 16384 2116; timeout -k 10 100 python tools/wave_times.py 4096; timeout -k 10 100 python tools/wave_times.py 8190 1053 drain; } > $O/wave_times_balance.txt 2>&1
grep -E "^==|in flight|XCD" $O/wave_times_balance.txt | grep -E "launch 2|in flight|XCD" | head -60

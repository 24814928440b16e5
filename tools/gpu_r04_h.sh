#!/bin/bash
# round 4: relay / marching break-even after the marching kernel's gains; edge class of the balance
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_clamped_step.py tests/test_hip_parity.py tests/test_full_size_golden.py tests/test_dry_tiles.py -m gpu -x -q > $O/pytest_h.log 2>&1 || { tail -n 30 $O/pytest_h.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_h.log)"
SHAPES="2000x2000:drain 2200x2200:drain 2400x2400:drain 2700x2700:drain 3000x3000:drain 1053x8190:drain 2000x2000:add 2200x2200:add 2400x2400:add 2700x2700:add 3000x3000:add" timeout -k 10 1000 python tools/ab_shapes.py 2 base "base WDPM_RELAY=0" > $O/relay_breakeven.txt 2>&1; cat $O/relay_breakeven.txt
SHAPES="16384x16384:add 4096x4096:add 2116x16384:add 8192x8192:drain" timeout -k 10 1000 python tools/ab_shapes.py 3 r3 "base WDPM_BALANCE=0" base > $O/balance_edge_ab.txt 2>&1; cat $O/balance_edge_ab.txt
export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so
{ timeout -k 10 200 python tools/wave_times.py 16384; timeout -k 10 100 python tools/wave_times.py 16384 2116; timeout -k 10 100 python tools/wave_times.py 4096; } > $O/wave_times_edge.txt 2>&1
grep -E "^==|in flight|last to end" $O/wave_times_edge.txt

#!/bin/bash
# round 4: cooperative issue priorities (eight-wave workgroups, progress through LDS) - kernel suites, A/B, per-wave timestamps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_clamped_step.py tests/test_hip_parity.py tests/test_full_size_golden.py tests/test_dry_tiles.py -m gpu -x -q > $O/pytest_f.log 2>&1 || { tail -n 30 $O/pytest_f.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_f.log)"
SHAPES="16384x16384:add 4096x4096:add 2116x16384:add 8192x8192:add 8192x8192:drain 1053x8190:drain 4096x4096:drain 3000x3000:add 6000x6000:add" timeout -k 10 1000 python tools/ab_shapes.py 3 r3 c1 base "base WDPM_PRIO=0" > $O/coop_prio_shapes_ab.txt 2>&1; cat $O/coop_prio_shapes_ab.txt
export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so
{ timeout -k 10 200 python tools/wave_times.py 16384; timeout -k 10 100 python tools/wave_times.py 16384 2116; timeout -k 10 100 python tools/wave_times.py 4096; timeout -k 10 100 python tools/wave_times.py 8190 1053 drain; } > $O/wave_times_coop.txt 2>&1
grep -E "^==|in flight|median end by slot" $O/wave_times_coop.txt

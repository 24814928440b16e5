#!/bin/bash
# round 4: saddr asm stores (real nt), NaN decode, OR-ed nzmask - kernel suites, then A/B: r3 / c1 (clamp only) / base, stores forced both ways
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_clamped_step.py tests/test_hip_parity.py tests/test_full_size_golden.py tests/test_dry_tiles.py -m gpu -x -q > $O/pytest_d.log 2>&1 || { tail -n 30 $O/pytest_d.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_d.log)"
SHAPES="16384x16384:add 4096x4096:add 2116x16384:add 8192x8192:add 8192x8192:drain 1053x8190:drain 4096x4096:drain 3000x3000:add" timeout -k 10 1000 python tools/ab_shapes.py 3 r3 c1 "base WDPM_STORES=plain" "base WDPM_STORES=nt" > $O/stores_shapes_ab.txt 2>&1; cat $O/stores_shapes_ab.txt

#!/bin/bash
# round 3, session L: the relay kernel for drain - parity (drain suites, every outlet position, the command line against the reference's reports), then small rasters
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_cli.py tests/test_cli_differential.py tests/test_rowblock.py tests/test_setup_stats.py -m gpu -x -q > $O/pytest_relay_drain.log 2>&1; rc=$?; echo "relay on: $(tail -n 1 $O/pytest_relay_drain.log)"
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_relay_drain.log | head -40; exit 1; }
for n in 300 482 700 1000; do python tools/drain_outlet_cost.py $n 3000; WDPM_RELAY=0 python tools/drain_outlet_cost.py $n 3000 | sed 's/^/WDPM_RELAY=0 /'; done 2>&1 | grep -v amdgpu.ids | tee $O/relay_drain.txt

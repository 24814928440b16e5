#!/bin/bash
# round 3, session M: the whole GPU suite with the relay kernel in place; the kernel suites with it off (WDPM_RELAY=0: the triangle kernel) and with the gated variants only
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest.log | head -40; exit 1; }
WDPM_RELAY=0 timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_rowblock.py -m gpu -x -q > $O/pytest_norelay.log 2>&1; rc=$?; echo "WDPM_RELAY=0: $(tail -n 1 $O/pytest_norelay.log)"
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_norelay.log | head -40; exit 1; }
WDPM_PLAIN=0 timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_rowblock.py -m gpu -x -q -k "not water_kinds" > $O/pytest_gated.log 2>&1; rc=$?; echo "WDPM_PLAIN=0: $(tail -n 1 $O/pytest_gated.log)"
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_gated.log | head -40; exit 1; }
echo "== CLI basin5"; timeout -k 10 600 python -m pytest tests/test_cli.py -m gpu -q -s -k "convergence" 2>&1 | grep -E "wall|passed|failed" | tee $O/cli_basin5.txt

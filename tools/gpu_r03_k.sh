#!/bin/bash
# round 3, session K: the relay kernel - parity with it (default) and without (WDPM_RELAY=0), then small rasters both ways
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_cli.py tests/test_cli_differential.py tests/test_rowblock.py -m gpu -x -q > $O/pytest_relay.log 2>&1; rc=$?; echo "relay on: $(tail -n 1 $O/pytest_relay.log)"
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_relay.log | head -40; exit 1; }
WDPM_RELAY=0 timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py -m gpu -x -q > $O/pytest_norelay.log 2>&1; rc=$?; echo "WDPM_RELAY=0: $(tail -n 1 $O/pytest_norelay.log)"
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_norelay.log | head -40; exit 1; }
for rep in 1 2; do for relay in 1 0 2; do for sz in 200 300 482 600; do echo -n "WDPM_RELAY=$relay add $sz: "; WDPM_RELAY=$relay timeout -k 10 200 python bench.py --size $sz --steps 3000 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us per iteration' % (d['ms_per_step']*1e3))"; done; done; done | tee $O/relay_ab.txt

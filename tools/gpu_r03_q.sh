#!/bin/bash
# round 3, session Q: relay kernel with the DEM as 32-bit codes in launches of several rounds (add): parity, A/B against the fp64 DEM, default dispatch against WDPM_RELAY=0
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_rowblock.py tests/test_cli.py -m gpu -x -q > $O/pytest_relay32.log 2>&1; rc=$?; echo "parity: $(tail -n 1 $O/pytest_relay32.log)"
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_relay32.log | head -40; exit 1; }
WDPM_RELAY=2 WDPM_RELAY_NW=8 timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -1
for rep in 1 2; do for sz in 700 1200 1600 2000 2400 3000 3600; do for cfg in "WDPM_RELAY=2 WDPM_RELAY_NW=8 WDPM_RELAY_DEM32=1" "WDPM_RELAY=2 WDPM_RELAY_NW=8 WDPM_RELAY_DEM32=0" "WDPM_RELAY=0 X=0 Y=0"; do
    steps=$(( 300000000 / (sz * sz / 100 + 10000) )); [ $steps -gt 3000 ] && steps=3000
    echo -n "add $sz $cfg: "; env $cfg timeout -k 10 200 python bench.py --size $sz --steps $steps --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us per iteration' % (d['ms_per_step']*1e3))"; done; done; done | tee $O/relay_dem32_ab.txt

#!/bin/bash
# round 3, session N: relay dispatch final - parity, default choices across sizes, the 8-GPU drain slab
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_cli.py tests/test_rowblock.py tests/test_full_size_golden.py tests/test_mock_rccl.py -m gpu -x -q -k "not standin" > $O/pytest_relay_final.log 2>&1; rc=$?; echo "parity: $(tail -n 1 $O/pytest_relay_final.log)"
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_relay_final.log | head -40; exit 1; }
for m in add drain; do for sz in 300 482 700 1000 1200 1600 2000 2400 3000 3600; do for relay in 1 0; do
  steps=$(( 300000000 / (sz * sz / 100 + 10000) )); [ $steps -gt 3000 ] && steps=3000
  echo -n "$m $sz WDPM_RELAY=$relay: "; WDPM_RELAY=$relay timeout -k 10 200 python bench.py --module $m --size $sz --steps $steps --warmup 50 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us per iteration' % (d['ms_per_step']*1e3))"; done; done; done 2>&1 | tee $O/relay_default.txt
echo "== slabs" | tee $O/slabs_relay.txt
for relay in 1 0; do echo "WDPM_RELAY=$relay"; WDPM_RELAY=$relay timeout -k 10 200 python tools/shape_bench.py 1055 8190 500 fused drain 2>/dev/null; WDPM_RELAY=$relay timeout -k 10 200 python tools/shape_bench.py 2049 16384 300 fused add 2>/dev/null; done | tee -a $O/slabs_relay.txt

#!/bin/bash
# round 3, session P: relay kernel, the taken-over rows' elevations through LDS (shipped) against five rows of DEM fetched per wave (alt_base)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_rowblock.py tests/test_cli.py -m gpu -x -q > $O/pytest_ldsdem.log 2>&1; rc=$?; echo "parity: $(tail -n 1 $O/pytest_ldsdem.log)"
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_ldsdem.log | head -40; exit 1; }
for rep in 1 2; do for v in new base; do
  if [ $v = base ]; then export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_base_libwdpm_hip.so; else unset WDPM_HIP_LIB; fi
  for m in add drain; do for sz in 482 700 1000 1200 1600 2000 2400 3000; do
    steps=$(( 300000000 / (sz * sz / 100 + 10000) )); [ $steps -gt 3000 ] && steps=3000
    echo -n "$v $m $sz: "; WDPM_RELAY=2 timeout -k 10 200 python bench.py --module $m --size $sz --steps $steps --warmup 50 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us per iteration' % (d['ms_per_step']*1e3))"; done; done
done; done | tee $O/relay_ldsdem_ab.txt

#!/bin/bash
# round 3, session D: dead warm-up stages left out - parity on the whole kernel suite, then A/B against the previous build over sizes and slab shapes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_full_size_golden.py tests/test_rowblock.py tests/test_stencil_forms.py -m gpu -x -q -k "not standin" > $O/pytest_warm.log 2>&1; rc=$?; tail -n 3 $O/pytest_warm.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_warm.log | head -40; exit 1; }
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  %.4g cell-updates/s  %.2f us/iteration' % (d['value'], d['ms_per_step']*1e3))"; }
for rep in 1 2; do for v in base new; do
  if [ $v = base ]; then export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_base_libwdpm_hip.so; else unset WDPM_HIP_LIB; fi
  for n in 2048 3072 4096 8192; do steps=$(( 400000000 / (n * n / 1000 + 1000) )); [ $steps -gt 2000 ] && steps=2000
    echo -n "$v add $n: "; timeout -k 10 200 python bench.py --size $n --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | line; done
  echo -n "$v add 16384: "; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v drain 4096: "; timeout -k 10 200 python bench.py --module drain --size 4096 --steps 500 --warmup 5 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v drain 8192: "; timeout -k 10 200 python bench.py --module drain --size 8192 --steps 300 --warmup 5 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v add 16384 on 8 slabs (group, one GPU): "; timeout -k 10 200 python bench.py --gpus 8 --driver group --steps 100 --warmup 8 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v drain 8192 on 8 slabs (group, one GPU): "; timeout -k 10 200 python bench.py --gpus 8 --driver group --module drain --size 8192 --steps 200 --warmup 8 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | line
done; done 2>&1 | tee $O/warmup_stages_ab.txt

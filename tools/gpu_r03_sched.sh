#!/bin/bash
# round 3: the marching kernel's translation unit under the compiler's max-ILP scheduler (shipped), against the small-raster
# kernels' unit under it too (allilp) and neither (noilp): parity on the shipped library, then A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_rowblock.py tests/test_full_size_golden.py -m gpu -x -q -k "not standin" > $O/pytest_sched.log 2>&1 || { tail -n 30 $O/pytest_sched.log; exit 1; }
echo "parity shipped: $(tail -n 1 $O/pytest_sched.log)"
WDPM_PRIO=2 WDPM_RELAY=0 WDPM_TRI=0 timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py -m gpu -x -q > $O/pytest_sched_marching.log 2>&1 || { tail -n 30 $O/pytest_sched_marching.log; exit 1; }
echo "parity shipped, PRIO forced, marching kernel only: $(tail -n 1 $O/pytest_sched_marching.log)"
us() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; }
for rep in 1 2; do for v in noilp shipped allilp; do
  if [ $v = shipped ]; then unset WDPM_HIP_LIB; else export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so; fi
  for m in add drain; do echo -n "$v $m us/iteration at 482 700 1000 1200 1600 2000 2400 3000 4096: "
    for n in 482 700 1000 1200 1600 2000 2400 3000 4096; do timeout -k 10 200 python bench.py --module $m --size $n --steps 2000 --warmup 50 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | us; done; echo; done
  echo -n "$v us/iteration add 8192, add 16384, drain 8192, slab add, slab drain: "
  timeout -k 10 200 python bench.py --size 8192 --steps 1000 --warmup 20 --no-cpu-baseline 2>/dev/null | us
  timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | us
  timeout -k 10 200 python bench.py --module drain --size 8192 --steps 300 --warmup 5 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | us
  timeout -k 10 200 python tools/shape_bench.py 2049 16384 300 fused add 2>/dev/null | awk '{printf "%s ", $(NF-1)}'
  timeout -k 10 200 python tools/shape_bench.py 1055 8190 500 fused drain 2>/dev/null | awk '{printf "%s ", $(NF-1)}'; echo
done; done 2>&1 | tee $O/sched_strategy_ab.txt

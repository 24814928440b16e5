#!/bin/bash
# round 3, session T: the PRIO instantiations of the marching kernel: kernel suites with them forced on every marching launch
# (with and without the small-raster kernels), wave timing as shipped, whole suite, A/B against WDPM_PRIO=0 (same library)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
WDPM_PRIO=2 timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_full_size_golden.py tests/test_rowblock.py tests/test_setup_stats.py -m gpu -x -q -k "not standin" > $O/pytest_prio_forced.log 2>&1 || { tail -n 30 $O/pytest_prio_forced.log; exit 1; }
echo "PRIO forced: $(tail -n 1 $O/pytest_prio_forced.log)"
WDPM_PRIO=2 WDPM_RELAY=0 WDPM_TRI=0 timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py -m gpu -x -q > $O/pytest_prio_forced_marching.log 2>&1 || { tail -n 30 $O/pytest_prio_forced_marching.log; exit 1; }
echo "PRIO forced, marching kernel only: $(tail -n 1 $O/pytest_prio_forced_marching.log)"
( WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so timeout -k 10 200 python tools/wave_times.py 16384 &&
  WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so timeout -k 10 200 python tools/wave_times.py 8192 8192 drain ) > $O/wave_times_prio.txt 2>&1
grep -A4 "launch 2" $O/wave_times_prio.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -n 30 $O/pytest.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest.log)"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  %.4g cell-updates/s  %.2f us/iteration  kernel %.4f ms' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_iteration']))"; }
for rep in 1 2; do for v in 0 1; do
  export WDPM_PRIO=$v
  for n in 4096 6000 7000 8192; do steps=$(( 400000000 / (n * n / 1000 + 1000) )); [ $steps -gt 2000 ] && steps=2000
    echo -n "WDPM_PRIO=$v add $n: "; timeout -k 10 200 python bench.py --size $n --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | line; done
  echo -n "WDPM_PRIO=$v add 16384: "; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "WDPM_PRIO=$v add 16384, 20 steps: "; timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line
  echo -n "WDPM_PRIO=$v slab add 8 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 2049 16384 300 fused add 2>/dev/null
  echo -n "WDPM_PRIO=$v slab add 4 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 4097 16384 200 fused add 2>/dev/null
  echo -n "WDPM_PRIO=$v slab add 2 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 8193 16384 100 fused add 2>/dev/null
  echo -n "WDPM_PRIO=$v slab drain 8 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 1055 8190 500 fused drain 2>/dev/null
  echo -n "WDPM_PRIO=$v slab drain 2 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 4097 8190 300 fused drain 2>/dev/null
  for n in 4096 6000 8192; do echo -n "WDPM_PRIO=$v drain $n: "; timeout -k 10 200 python bench.py --module drain --size $n --steps 300 --warmup 5 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | line; done
done; done 2>&1 | tee $O/prio_ab.txt

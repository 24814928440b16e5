#!/bin/bash
# round 3: three waves per SIMD for the add / subtract instances that stream the DEM as codes (168 VGPRs, no scratch):
# parity on the variant, then A/B against the shipped build on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  %.4g cell-updates/s  %.2f us/iteration  kernel %.4f ms' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_iteration']))"; }
export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_w3_libwdpm_hip.so
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_full_size_golden.py -m gpu -x -q -k "not standin" > $O/pytest_w3.log 2>&1 || { tail -n 30 $O/pytest_w3.log; exit 1; }
echo "parity w3: $(tail -n 1 $O/pytest_w3.log)"
for rep in 1 2; do for v in base w3; do
  export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so
  for n in 4096 6000 8192; do steps=$(( 400000000 / (n * n / 1000 + 1000) )); [ $steps -gt 2000 ] && steps=2000
    echo -n "$v add $n: "; timeout -k 10 200 python bench.py --size $n --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | line; done
  echo -n "$v add 16384: "; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v slab: "; timeout -k 10 200 python tools/shape_bench.py 2049 16384 300 fused add 2>/dev/null
done; done 2>&1 | tee $O/waves3_ab.txt

#!/bin/bash
# round 3, session Y: the add PRIO instantiations without the dead stages of a chunk's first two steps: parity, A/B against -DWDPM_PRIO_PEEL=0
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
WDPM_PRIO=2 WDPM_RELAY=0 WDPM_TRI=0 timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py -m gpu -x -q > $O/pytest_peel_marching.log 2>&1 || { tail -n 30 $O/pytest_peel_marching.log; exit 1; }
echo "peel, PRIO forced, marching kernel only: $(tail -n 1 $O/pytest_peel_marching.log)"
timeout -k 10 900 python -m pytest tests/test_full_size_golden.py tests/test_rowblock.py -m gpu -x -q -k "not standin" > $O/pytest_peel.log 2>&1 || { tail -n 30 $O/pytest_peel.log; exit 1; }
echo "peel, full size + row blocks: $(tail -n 1 $O/pytest_peel.log)"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  %.4g cell-updates/s  %.2f us/iteration  kernel %.4f ms' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_iteration']))"; }
for rep in 1 2 3; do for v in nopeel shipped; do
  if [ $v = shipped ]; then unset WDPM_HIP_LIB; else export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so; fi
  echo -n "$v add 6000: "; timeout -k 10 200 python bench.py --size 6000 --steps 1000 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v add 8192: "; timeout -k 10 200 python bench.py --size 8192 --steps 1000 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v add 16384: "; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v slab add 8 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 2049 16384 300 fused add 2>/dev/null
  echo -n "$v slab add 4 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 4097 16384 200 fused add 2>/dev/null
done; done 2>&1 | tee $O/prio_peel_ab.txt

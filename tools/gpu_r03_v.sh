#!/bin/bash
# round 3, session V: where do the PRIO instantiations start to pay?  WDPM_PRIO=0 (never) against 2 (always), same library
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  %.4g cell-updates/s  %.2f us/iteration  kernel %.4f ms' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_iteration']))"; }
for rep in 1 2 3; do for v in 0 2; do
  export WDPM_PRIO=$v
  echo -n "WDPM_PRIO=$v slab add 8 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 2049 16384 300 fused add 2>/dev/null
  echo -n "WDPM_PRIO=$v slab drain 8 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 1055 8190 500 fused drain 2>/dev/null
  echo -n "WDPM_PRIO=$v slab drain 4 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 2079 8190 500 fused drain 2>/dev/null
  for n in 4096 5000 6000; do steps=$(( 400000000 / (n * n / 1000 + 1000) )); [ $steps -gt 2000 ] && steps=2000
    echo -n "WDPM_PRIO=$v add $n: "; timeout -k 10 200 python bench.py --size $n --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | line; done
  for n in 4096 5000; do echo -n "WDPM_PRIO=$v drain $n: "; timeout -k 10 200 python bench.py --module drain --size $n --steps 500 --warmup 5 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | line; done
done; done 2>&1 | tee $O/prio_threshold_ab.txt

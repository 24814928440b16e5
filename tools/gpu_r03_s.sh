#!/bin/bash
# round 3, session S: A/B of the shipped library against -DWDPM_PRIO=0 on the shapes with short chunks
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  %.4g cell-updates/s  %.2f us/iteration  kernel %.4f ms' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_iteration']))"; }
for rep in 1 2; do for v in noprio shipped; do
  if [ $v = shipped ]; then unset WDPM_HIP_LIB; else export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so; fi
  for n in 3072 4096 5000 6000 7000; do steps=$(( 400000000 / (n * n / 1000 + 1000) )); [ $steps -gt 2000 ] && steps=2000
    echo -n "$v add $n: "; timeout -k 10 200 python bench.py --size $n --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | line; done
  echo -n "$v add 16384: "; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v slab add: "; timeout -k 10 200 python tools/shape_bench.py 2049 16384 300 fused add 2>/dev/null
  echo -n "$v slab add 4 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 4097 16384 200 fused add 2>/dev/null
  echo -n "$v slab drain: "; timeout -k 10 200 python tools/shape_bench.py 1055 8190 500 fused drain 2>/dev/null
  echo -n "$v slab drain 4 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 2079 8190 500 fused drain 2>/dev/null
  for n in 3072 4096 6000; do echo -n "$v drain $n: "; timeout -k 10 200 python bench.py --module drain --size $n --steps 500 --warmup 5 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | line; done
done; done 2>&1 | tee $O/prio_short_ab.txt

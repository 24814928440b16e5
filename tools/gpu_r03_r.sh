#!/bin/bash
# round 3, session R: the progress-priority marching kernel as shipped: whole -m gpu suite, A/B against -DWDPM_PRIO=0 on this box,
# then the evidence session (tools/gpu_r03_final.sh) and the add command's rocprofv3 passes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -n 30 $O/pytest.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest.log)"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  %.4g cell-updates/s  %.2f us/iteration  kernel %.4f ms' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_iteration']))"; }
for rep in 1 2; do for v in noprio shipped; do
  if [ $v = shipped ]; then unset WDPM_HIP_LIB; else export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so; fi
  for n in 4096 6000 8192; do steps=$(( 400000000 / (n * n / 1000 + 1000) )); [ $steps -gt 2000 ] && steps=2000
    echo -n "$v add $n: "; timeout -k 10 200 python bench.py --size $n --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | line; done
  echo -n "$v add 16384: "; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v add 16384, 20 steps: "; timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v slab add: "; timeout -k 10 200 python tools/shape_bench.py 2049 16384 300 fused add 2>/dev/null
  echo -n "$v slab drain: "; timeout -k 10 200 python tools/shape_bench.py 1055 8190 500 fused drain 2>/dev/null
  echo -n "$v drain 4096: "; timeout -k 10 200 python bench.py --module drain --size 4096 --steps 500 --warmup 5 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | line
  echo -n "$v drain 8192: "; timeout -k 10 200 python bench.py --module drain --size 8192 --steps 300 --warmup 5 --drain-spinup 100 --no-cpu-baseline 2>/dev/null | line
done; done 2>&1 | tee $O/prio_ab.txt
unset WDPM_HIP_LIB
bash tools/gpu_r03_final.sh && bash tools/profile.sh r03/bench16k > $O/bench16k_profile.txt 2>&1; grep -h '"metric"' $O/bench16k/trace.log | cut -c1-200

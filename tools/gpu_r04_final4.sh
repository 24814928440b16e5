#!/bin/bash
# round 4, after the drain kernel got the DEM codes: PMC passes again (profiles/traffic.json names the build), drain 8192^2 traced and
# counted, the bench lines of configs 3 - 5, the slabs, the driver's command
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04/final4; mkdir -p $O
bash tools/profile.sh r04/final4/prof16k > $O/profile16k.txt 2>&1; tail -n 2 $O/profile16k.txt | cut -c1-100
timeout -k 10 300 bash tools/profile.sh r04/final4/profdrain8192 --module drain --size 8192 --drain-spinup 100 > $O/profiledrain.txt 2>&1; grep -h fused $O/profdrain8192/trace/*/*_kernel_stats.csv | cut -c1-60,330-420 | head -3
for a in "--steps 1000 --warmup 20:bench_default" "--size 4096 --steps 1000 --warmup 20:config3_4096" "--module drain --size 8192 --steps 1000 --warmup 20 --drain-spinup 1000:config5_drain_8192_1gpu" "--module drain --size 8192 --steps 1000 --warmup 300 --drain-spinup 1000:config5_drain_8192_1gpu_warm300"; do
  args=${a%%:*}; name=${a##*:}
  timeout -k 10 400 python bench.py $args --no-cpu-baseline > $O/$name.json 2> $O/$name.err || echo "$name failed"
  python -c "import json,sys; d=json.load(open('$O/$name.json')); r=d['roofline']; print('$name: value %.4g  ms/step %.4f  kernel_ms %.4f  frac %.3f  job_frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_per_iteration'], r['frac'], r['job_frac']))"
done
{ echo "== slabs of the 8-GPU runs, each alone on the GPU"; python tools/shape_bench.py 2049 16384 200 fused add; python tools/shape_bench.py 1055 8190 400 fused drain; WDPM_DEM32=0 python tools/shape_bench.py 1055 8190 400 fused drain; python tools/shape_bench.py 1055 8190 400 fused drain; WDPM_DEM32=0 python tools/shape_bench.py 1055 8190 400 fused drain; } > $O/slabs.txt 2>&1; grep us/iter $O/slabs.txt

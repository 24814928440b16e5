#!/bin/bash
# (second half of the evidence session: see gpu_r04_final.sh)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04/final; mkdir -p $O
bash tools/profile.sh r04/final/prof16k > $O/profile16k.txt 2>&1; tail -n 3 $O/profile16k.txt | cut -c1-200
for a in "--steps 20 --warmup 5:bench_steps20" "--steps 1000 --warmup 20:bench_default" "--size 4096 --steps 1000 --warmup 20:config3_4096" "--module drain --size 8192 --steps 1000 --warmup 20 --drain-spinup 1000:config5_drain_8192_1gpu"; do
  args=${a%%:*}; name=${a##*:}
  timeout -k 10 400 python bench.py $args $( [ $name = bench_steps20 ] || echo --no-cpu-baseline ) > $O/$name.json 2> $O/$name.err || echo "$name failed"
  python -c "import json,sys; d=json.load(open('$O/$name.json')); r=d['roofline']; print('$name: value %.4g  ms/step %.4f  kernel_ms %.4f  frac %.3f  job_frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_per_iteration'], r['frac'], r['job_frac']))"
done
{ echo "--- k = 8, overlapped last iteration"; timeout -k 10 400 python tools/scale_projection.py 16384 8 40; } > $O/scale_projection.txt 2>&1; grep "^N=" $O/scale_projection.txt | cut -c1-170
{ echo "== slabs of the 8-GPU runs, each alone on the GPU"; python tools/shape_bench.py 2049 16384 200 fused add; python tools/shape_bench.py 1055 8190 400 fused drain; } > $O/slabs.txt 2>&1; grep us/iter $O/slabs.txt
timeout -k 10 300 bash tools/profile.sh r04/final/prof4096 --size 4096 > $O/profile4096.txt 2>&1; grep -h fused $O/prof4096/trace/*/*_kernel_stats.csv | cut -c1-60,330-420 | head -3
timeout -k 10 300 bash tools/profile.sh r04/final/profdrain8192 --module drain --size 8192 --drain-spinup 100 > $O/profiledrain.txt 2>&1; grep -h fused $O/profdrain8192/trace/*/*_kernel_stats.csv | cut -c1-60,330-420 | head -3

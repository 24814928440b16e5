#!/bin/bash
# round 4: evidence for profiles/r04 - rocprofv3 kernel trace + PMC passes of the bench command, bench lines of the configs, scale projection, slabs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
bash tools/profile.sh r04/prof16k > $O/profile16k.txt 2>&1; tail -n 12 $O/profile16k.txt | cut -c1-300
for a in "--steps 20 --warmup 5:bench_steps20" "--steps 1000 --warmup 20:bench_default" "--size 4096 --steps 1000 --warmup 20:config3_4096" "--module drain --size 8192 --steps 1000 --warmup 20 --drain-spinup 1000:config5_drain_8192_1gpu"; do
  args=${a%%:*}; name=${a##*:}
  timeout -k 10 300 python bench.py $args --no-cpu-baseline > $O/$name.json 2> $O/$name.err || echo "$name failed"
  python -c "import json,sys; d=json.load(open('$O/$name.json')); r=d['roofline']; print('$name: value %.4g  ms/step %.4f  kernel_ms %.4f  frac %.3f  job_frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_per_iteration'], r['frac'], r['job_frac']))"
done
{ echo "--- k = 8, overlapped last iteration"; timeout -k 10 400 python tools/scale_projection.py 16384 8 40; } > $O/scale_projection.txt 2>&1; cat $O/scale_projection.txt | cut -c1-200
{ echo "== slabs of the 8-GPU runs, each alone on the GPU"; python tools/shape_bench.py 2049 16384 200 fused add; python tools/shape_bench.py 1055 8190 400 fused drain; } > $O/slabs.txt 2>&1; cat $O/slabs.txt

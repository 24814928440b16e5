#!/bin/bash
# The evidence session: regenerates every file profiles/rNN/README.md cites, on ONE GPU box.  Parts are named so that a session can be
# split over gpurun calls (each at most 20 minutes):
#     gpurun --timeout 1200 -- 'bash tools/evidence_session.sh r05 suite'
#     gpurun --timeout 1200 -- 'bash tools/evidence_session.sh r05 profile lines'
#     gpurun --timeout 1200 -- 'bash tools/evidence_session.sh r05 scale slabs sweep e2e'
# Output: gpurun_out/<round>/final/; copy what is to be judged into profiles/<round>/ (tools/make_traffic_json.py writes
# profiles/traffic.json from the PMC passes).  Replaces round 4's tools/gpu_r04_final[1-4].sh.
#   suite    the whole `-m gpu` test suite
#   profile  rocprofv3 --kernel-trace --stats + the separate PMC passes of the bench command at 16384^2, 4096^2, drain 8192^2 (tools/profile.sh)
#   lines    bench.py lines: the driver's command (--steps 20 --warmup 5, with the CPU baseline), the default 1000-iteration block,
#            BASELINE config 3 (4096^2) and config 5 on one GPU (drain 8192^2)
#   scale    the middle slab of an N-GPU run alone on the GPU, queued as the driver queues it (tools/scale_projection.py)
#   slabs    the 8-GPU slabs of configs 4 and 5 alone on the GPU (tools/shape_bench.py)
#   sweep    the driver's command at --warmup 5 / 30 / 100 (the clock ramp)
#   e2e      WDPMCL end to end at 16384^2 (tools/e2e_16k.sh)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}; R=${1:?round, e.g. r05}; shift; O=gpurun_out/$R/final; mkdir -p $O
line() { python -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2: value %.4g  ms/step %.4f  kernel_ms %.4f  frac %.3f  job_frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_per_iteration'], r['frac'], r['job_frac']))"; }
for part in "$@"; do case $part in
suite)
  timeout -k 10 1150 python -m pytest tests -m gpu -x -q -rs > $O/pytest_gpu.log 2>&1 || { tail -n 30 $O/pytest_gpu.log; exit 1; }
  echo "suite: $(tail -n 1 $O/pytest_gpu.log)" ;;
profile)
  bash tools/profile.sh $R/final/prof16k > $O/profile16k.txt 2>&1 && tail -n 3 $O/profile16k.txt | cut -c1-200 &&
  timeout -k 10 300 bash tools/profile.sh $R/final/prof4096 --size 4096 > $O/profile4096.txt 2>&1 && grep -h fused $O/prof4096/trace/*/*_kernel_stats.csv | cut -c1-60,330-420 | head -3 &&
  timeout -k 10 300 bash tools/profile.sh $R/final/profdrain8192 --module drain --size 8192 --drain-spinup 100 > $O/profiledrain.txt 2>&1 && grep -h fused $O/profdrain8192/trace/*/*_kernel_stats.csv | cut -c1-60,330-420 | head -3 || exit 1 ;;
lines)
  for a in "--steps 20 --warmup 5:bench_steps20" "--steps 1000 --warmup 20:bench_default" "--size 4096 --steps 1000 --warmup 20:config3_4096" "--module drain --size 8192 --steps 1000 --warmup 20 --drain-spinup 1000:config5_drain_8192_1gpu"; do
    args=${a%%:*}; name=${a##*:}
    timeout -k 10 400 python bench.py $args $( [ $name = bench_steps20 ] || echo --no-cpu-baseline ) > $O/$name.json 2> $O/$name.err || { echo "$name failed"; exit 1; }
    line $O/$name.json $name
  done ;;
scale)
  { echo "--- k = 8, overlapped last iteration"; timeout -k 10 400 python tools/scale_projection.py 16384 8 40; } > $O/scale_projection.txt 2>&1 || exit 1
  grep "^N=" $O/scale_projection.txt | cut -c1-170 ;;
slabs)
  { echo "== slabs of the 8-GPU runs, each alone on the GPU"; python tools/shape_bench.py 2049 16384 200 fused add && python tools/shape_bench.py 1055 8190 400 fused drain &&
    python tools/shape_bench.py 2049 16384 200 fused add && python tools/shape_bench.py 1055 8190 400 fused drain; } > $O/slabs.txt 2>&1 || exit 1
  grep us/iter $O/slabs.txt ;;
sweep)
  for w in 5 30 100; do echo -n "--steps 20 --warmup $w: "; timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline 2>/dev/null > $O/_w.json && line $O/_w.json "warmup $w" || exit 1; done > $O/warm_sweep.txt 2>&1
  cat $O/warm_sweep.txt ;;
e2e)
  bash tools/e2e_16k.sh > $O/e2e_cli_16384.txt 2>&1 || exit 1; cat $O/e2e_cli_16384.txt ;;
*) echo "unknown part $part"; exit 2 ;;
esac; done

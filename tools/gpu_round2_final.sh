#!/bin/bash
# Round-2 evidence session on one MI355X box: parity suite, rocprofv3 trace + PMC passes of the bench commands, bench lines,
# small rasters (triangle kernel A/B), sparse raster, 8 rank threads on one GPU, self-launched ranks, end-to-end CLI.
# Everything lands under gpurun_out/r02/ ; the summaries judged are copied to profiles/r02/ afterwards.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest.log | head -30; exit 1; }
bash tools/profile.sh r02/bench16k > $O/bench16k_profile.txt 2>&1; grep -h '"metric"' $O/bench16k/trace.log | cut -c1-160
bash tools/profile.sh r02/drain8192 --module drain --size 8192 --drain-spinup 2 > $O/drain8192_profile.txt 2>&1; grep -h '"metric"' $O/drain8192/trace.log | cut -c1-160
bash tools/profile.sh r02/add482 --size 482 > $O/add482_profile.txt 2>&1; grep -h '"metric"' $O/add482/trace.log | cut -c1-160
echo "== bench default"; timeout -k 10 300 python bench.py > $O/bench_default.json 2>/dev/null; cut -c1-200 $O/bench_default.json
echo "== bench --steps 20 --warmup 5"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_steps20.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['job_frac'])"
echo "== fp64 DEM"; WDPM_DEM32=0 timeout -k 10 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_fp64dem.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_fp64dem.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
echo "== drain 8192"; timeout -k 10 300 python bench.py --module drain --size 8192 --steps 1000 --warmup 5 --drain-spinup 200 --no-cpu-baseline > $O/config5_drain_8192_1gpu.json 2>/dev/null; cut -c1-160 $O/config5_drain_8192_1gpu.json
echo "== config 3: 4096"; timeout -k 10 300 python bench.py --size 4096 --steps 1000 --warmup 20 --no-cpu-baseline > $O/config3_4096.json 2>/dev/null; cut -c1-160 $O/config3_4096.json
echo "== small rasters, triangle kernel on / off" | tee $O/tri_ab.txt
for tri in 1 0; do for sz in 482 700 1000; do for m in add drain; do echo -n "WDPM_TRI=$tri $m $sz: "; WDPM_TRI=$tri timeout -k 10 300 python bench.py --module $m --size $sz --steps 3000 --warmup 50 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g cell-updates/s  %.2f us per iteration' % (d['value'], d['ms_per_step']*1e3))"; done; done; done | tee -a $O/tri_ab.txt
echo "== sparse raster" | tee $O/sparse.txt; for p in 0 1 12; do echo "ponds $p"; timeout -k 10 300 python tools/sparse_bench.py 16384 100 2 $p 2>&1 | grep -E "tiles=|identical"; done | tee -a $O/sparse.txt
echo "8192^2, 12 ponds" | tee -a $O/sparse.txt; timeout -k 10 300 python tools/sparse_bench.py 8192 100 2 12 2>&1 | grep -E "tiles=|identical" | tee -a $O/sparse.txt
echo "== 8 rank threads on ONE GPU (peer copies): host cost of queueing" | tee $O/group8_enqueue.txt
for a in "--size 16384 --steps 200 --warmup 8 --exchange-every 4" "--size 16384 --steps 200 --warmup 8 --exchange-every 8" "--size 2048 --steps 2000 --warmup 40 --exchange-every 4"; do timeout -k 10 300 python bench.py --gpus 8 --driver group $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$a: ms per step %.4f  queueing %.1f us per iteration and rank  halo refresh (host side, incl. waiting) %.1f us  halo %s' % (d['ms_per_step'], c['enqueue_us_per_iteration_per_rank'], c['halo_refresh_host_us_per_iteration_per_rank'], c['halo']))"; done | tee -a $O/group8_enqueue.txt
echo "== python bench.py --gpus 2 (starts its own ranks; one GPU here -> gloo + host halos)"; timeout -k 10 300 python bench.py --gpus 2 --size 8192 --steps 100 --warmup 8 > $O/bench_ranks2_one_gpu.json 2>/dev/null; cut -c1-700 $O/bench_ranks2_one_gpu.json
echo "== end to end through WDPMCL, 16384^2"; bash tools/e2e_16k.sh > $O/e2e_cli_16384.txt 2>&1; tail -n 12 $O/e2e_cli_16384.txt
echo "== CLI basin5"; timeout -k 10 600 python -m pytest tests/test_cli.py -m gpu -q -s -k "convergence" 2>&1 | grep -E "wall|passed|failed" | tee $O/cli_basin5.txt

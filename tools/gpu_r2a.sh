#!/bin/bash
# round-2 session A: parity suite, bench (1 GPU), self-launched ranks + in-process group rehearsals on one GPU,
# enqueue cost of 8 rank threads, small-raster timing breakdown
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r2a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 5 $O/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED" $O/pytest.log | head -20; exit 1; }
echo "== bench default"; timeout -k 10 300 python bench.py > $O/bench_default.json 2>$O/bench_default.err; cut -c1-600 $O/bench_default.json
echo "== bench --gpus 2 (self-launched, shared GPU -> gloo+host)"; timeout -k 10 300 python bench.py --gpus 2 --size 8192 --steps 100 --warmup 8 > $O/bench_ranks2.json 2>$O/bench_ranks2.err; cut -c1-700 $O/bench_ranks2.json
echo "== bench --driver group --gpus 8 (8 slabs of one GPU, peer copies, thread per rank)"; timeout -k 10 300 python bench.py --gpus 8 --driver group --steps 200 --warmup 8 > $O/bench_group8.json 2>$O/bench_group8.err; cut -c1-900 $O/bench_group8.json
echo "== same, 1 slab"; timeout -k 10 300 python bench.py --steps 200 --warmup 8 --no-cpu-baseline > $O/bench_1.json 2>/dev/null; cut -c1-300 $O/bench_1.json
echo "== small raster (482): wall per iteration and kernel time"
timeout -k 10 120 python bench.py --size 482 --steps 3000 --warmup 100 --no-cpu-baseline > $O/bench_482.json 2>/dev/null; cut -c1-400 $O/bench_482.json
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof482 -- python3 $GRAFT_REPO_ROOT/bench.py --size 482 --steps 2000 --warmup 50 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof482.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof482 -name "*kernel_stats.csv" | head -1 | xargs -r head -8

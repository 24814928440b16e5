#!/bin/bash
# round 4: the DEM codes as 16-bit offsets, revisited on the round-4 kernel - parity, then A/B (same library, WDPM_DEM16=0/1) on bench and shapes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_forced_variants.py -m gpu -x -q > $O/pytest_n.log 2>&1 || { tail -n 30 $O/pytest_n.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_n.log)"
bash tools/ab_interleaved.sh 4 base "base WDPM_DEM16=0" > $O/dem16_bench_ab.txt 2>&1; tail -n 3 $O/dem16_bench_ab.txt
BENCH_ARGS="--steps 20 --warmup 5" bash tools/ab_interleaved.sh 3 base "base WDPM_DEM16=0" > $O/dem16_bench20_ab.txt 2>&1; tail -n 3 $O/dem16_bench20_ab.txt
BENCH_ARGS="--size 4096 --steps 200 --warmup 20" bash tools/ab_interleaved.sh 3 base "base WDPM_DEM16=0" > $O/dem16_bench4096_ab.txt 2>&1; tail -n 3 $O/dem16_bench4096_ab.txt
BENCH_ARGS="--size 8192 --steps 100 --warmup 20" bash tools/ab_interleaved.sh 3 base "base WDPM_DEM16=0" > $O/dem16_bench8192_ab.txt 2>&1; tail -n 3 $O/dem16_bench8192_ab.txt

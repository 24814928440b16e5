#!/bin/bash
# round 4: the window as a ring of nine row slots (no slide) - kernel suites, then interleaved A/B against the last commit
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_forced_variants.py -m gpu -x -q > $O/pytest_o.log 2>&1 || { tail -n 30 $O/pytest_o.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_o.log)"
bash tools/ab_interleaved.sh 4 c3 base > $O/ring_bench_ab.txt 2>&1; tail -n 2 $O/ring_bench_ab.txt
SHAPES="4096x4096:add 2116x16384:add 8192x8192:add 8192x8192:drain 1053x8190:drain 3000x3000:add" timeout -k 10 800 python tools/ab_shapes.py 3 c3 base > $O/ring_shapes_ab.txt 2>&1; cat $O/ring_shapes_ab.txt

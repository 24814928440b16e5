#!/bin/bash
# round 4: the drain step without its dead min (14 instructions): the whole suite, then A/B against the last commit
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_l.log 2>&1 || { tail -n 30 $O/pytest_l.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_l.log)"
SHAPES="8192x8192:drain 1053x8190:drain 4096x4096:drain 2000x2000:drain 482x471:drain 3000x3000:add" timeout -k 10 600 python tools/ab_shapes.py 3 c2 base > $O/drain14_ab.txt 2>&1; cat $O/drain14_ab.txt

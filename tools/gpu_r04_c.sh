#!/bin/bash
# round 4: the clamped neighbour step in the marching and relay kernels - whole -m gpu suite, then shapes A/B against round 3's library
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_c.log 2>&1 || { tail -n 30 $O/pytest_c.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_c.log)"
timeout -k 10 900 python tools/ab_shapes.py 3 r3 base > $O/clamp_shapes_ab.txt 2>&1; cat $O/clamp_shapes_ab.txt

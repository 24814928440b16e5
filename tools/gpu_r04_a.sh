#!/bin/bash
# round 4, first session: the clamped neighbour step in the marching kernel - parity, then A/B against round 3's library
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_clamped_step.py tests/test_hip_parity.py tests/test_full_size_golden.py -m gpu -x -q > $O/pytest_a.log 2>&1 || { tail -n 30 $O/pytest_a.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_a.log)"
bash tools/ab.sh r3 > $O/clamp_ab.txt 2>&1; cat $O/clamp_ab.txt

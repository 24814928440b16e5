#!/bin/bash
# round 5, session 2: the whole -m gpu suite on the new code (seats, slot filling + pairing, settled goldens) -> gpurun_out/r05/s2/
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s2; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -x -q -rs --durations=25 > $O/pytest_gpu.log 2>&1 || { tail -n 60 $O/pytest_gpu.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_gpu.log)"; grep -A 28 "slowest" $O/pytest_gpu.log | cut -c1-150

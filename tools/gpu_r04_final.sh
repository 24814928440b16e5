#!/bin/bash
# round 4, evidence session on the final code: the whole -m gpu suite, rocprofv3 kernel trace + PMC passes of the bench command, the bench
# lines of the BASELINE configs, scale projection, the 8-GPU slabs alone on a GPU, per-wave timestamps.  -> gpurun_out/r04/final/
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04/final; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -x -q -rs > $O/pytest_gpu.log 2>&1 || { tail -n 30 $O/pytest_gpu.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_gpu.log)"

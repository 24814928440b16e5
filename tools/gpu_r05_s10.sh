#!/bin/bash
# round 5, session 10: the whole -m gpu suite on the code with physical-XCD shares, the first-chunk class and the last strip on the interior code
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s10; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -q -rs --durations=12 > $O/pytest_gpu.log 2>&1; rc=$?
echo "suite (rc $rc): $(tail -n 1 $O/pytest_gpu.log)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20; grep -A 14 "slowest" $O/pytest_gpu.log | cut -c1-150
exit $rc

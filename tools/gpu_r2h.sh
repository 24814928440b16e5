#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r2h; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest.log | head -30; exit 1; }
echo "== end to end, 16384^2 through WDPMCL"; bash tools/e2e_16k.sh > $O/e2e_16k.txt 2>&1; tail -n 25 $O/e2e_16k.txt

#!/bin/bash
# round-2 session C: full parity suite (new full-size goldens), drain 8192 profile with per-template-instance counters
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r2c; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED" $O/pytest.log | head -20; exit 1; }
bash tools/profile.sh r2c/drain8192 --module drain --size 8192 --drain-spinup 2 > $O/drain8192_profile.txt 2>&1; tail -n 60 $O/drain8192_profile.txt

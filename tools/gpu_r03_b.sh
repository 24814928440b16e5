#!/bin/bash
# round 3, session B: whole GPU suite, then the round's baseline profile of the bench command
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest.log | head -40; exit 1; }
cat gpurun_out/mock_rccl_wire.txt
echo "== bench --steps 20 --warmup 5"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>$O/bench_steps20.err; cut -c1-2500 $O/bench_steps20.json

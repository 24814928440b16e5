#!/bin/bash
# round 3, session I: the outlet's block inside the triangle kernel's lockstep - parity of the drain paths, then what the outlet costs now
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_rowblock.py tests/test_cli.py tests/test_cli_differential.py -m gpu -x -q -k "drain or outlet or random or chain or differential or golden" > $O/pytest_outlet.log 2>&1; rc=$?; tail -n 3 $O/pytest_outlet.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_outlet.log | head -40; exit 1; }
for n in 482 700 1000; do python tools/drain_outlet_cost.py $n 3000; done 2>&1 | grep -v amdgpu.ids | tee $O/drain_outlet_cost.txt
for n in 1200 1600 2000; do python tools/drain_outlet_cost.py $n 1500; done 2>&1 | grep -v amdgpu.ids | tee -a $O/drain_outlet_cost.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r2m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_dry_tiles.py tests/test_hip_parity.py tests/test_rowblock.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest.log | head -30; exit 1; }
for h in 48 96 192; do echo "== sparse rows $h"; WDPM_SPARSE_ROWS=$h timeout -k 10 300 python tools/sparse_bench.py 16384 100 2 2>&1 | grep "tiles=1"; done

#!/bin/bash
cd $GRAFT_REPO_ROOT
for h in 24 48 96 192 384; do echo "== sparse rows $h"; WDPM_SPARSE_ROWS=$h timeout -k 10 300 python tools/sparse_bench.py 16384 100 2 2>&1 | grep "tiles=1"; done

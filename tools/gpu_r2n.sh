#!/bin/bash
cd $GRAFT_REPO_ROOT
for p in 0 1 12; do echo "== ponds $p"; timeout -k 10 300 python tools/sparse_bench.py 16384 100 2 $p 2>&1 | grep "tiles=1"; done
echo "== 8192, 12 ponds"; timeout -k 10 300 python tools/sparse_bench.py 8192 100 2 12 2>&1 | grep "tiles="

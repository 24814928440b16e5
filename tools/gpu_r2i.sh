#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r2i; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest.log | head -30; exit 1; }
echo "== dense bench (tiles on / off)"; for t in 1 0; do WDPM_TILES=$t timeout -k 10 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tiles=$t', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_iteration'])"; done
echo "== drain 8192 (tiles on / off)"; for t in 1 0; do WDPM_TILES=$t timeout -k 10 300 python bench.py --module drain --size 8192 --steps 300 --warmup 5 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tiles=$t', d['value'], d['ms_per_step'])"; done
echo "== sparse raster: 16384^2, 50 % NODATA, localized water"; timeout -k 10 600 python tools/sparse_bench.py 16384 2>&1 | tail -n 8

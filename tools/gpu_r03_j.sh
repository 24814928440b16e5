#!/bin/bash
# round 3, session J: the triangle kernel's max-diff variant: parity (block loops, the command line's per-block max diff against the reference's reports), then a 20-step block of a basin5-sized raster
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_cli.py tests/test_cli_differential.py tests/test_rowblock.py -m gpu -x -q > $O/pytest_trimd.log 2>&1; rc=$?; tail -n 3 $O/pytest_trimd.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_trimd.log | head -40; exit 1; }
for tri in 1 0; do for sz in 482 1000; do echo -n "WDPM_TRI=$tri add $sz, one block of 20 iterations: "; WDPM_TRI=$tri timeout -k 10 200 python bench.py --size $sz --steps 20 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us per iteration' % (d['ms_per_step']*1e3))"; done; done | tee $O/tri_md.txt

#!/bin/bash
# round 3, session H: the whole GPU suite; the kernel suites once more with the gated variants only (WDPM_PLAIN=0)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest.log | head -40; exit 1; }
WDPM_PLAIN=0 timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py tests/test_rowblock.py -m gpu -x -q -k "not water_kinds" > $O/pytest_gated.log 2>&1; rc=$?; echo "WDPM_PLAIN=0: $(tail -n 1 $O/pytest_gated.log)"
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_gated.log | head -40; exit 1; }
echo "== bench --steps 20 --warmup 5"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>$O/bench_steps20.err; cut -c1-400 $O/bench_steps20.json
echo "== config 5 on one GPU"; timeout -k 10 300 python bench.py --module drain --size 8192 --steps 1000 --warmup 5 --drain-spinup 200 --no-cpu-baseline > $O/config5_drain_8192_1gpu.json 2>/dev/null; cut -c1-300 $O/config5_drain_8192_1gpu.json

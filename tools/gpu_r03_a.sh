#!/bin/bash
# round 3, session A: the new N>1 tests (process-per-rank stand-in RCCL, deadlines, default-schedule goldens), then the whole suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_mock_rccl.py tests/test_bench_contract.py tests/test_rccl_transport.py -m gpu -x -q > $O/pytest_new.log 2>&1; rc=$?; tail -n 5 $O/pytest_new.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_new.log | head -40; exit 1; }
timeout -k 10 900 python -m pytest tests/test_full_size_golden.py "tests/test_cli.py::test_hip_cli_report_backend_line_and_gdaldem_handoff" -m gpu -x -q > $O/pytest_full.log 2>&1; rc=$?; tail -n 5 $O/pytest_full.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest_full.log | head -40; exit 1; }
echo "== bench --steps 20 --warmup 5"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>$O/bench_steps20.err; cut -c1-1500 $O/bench_steps20.json

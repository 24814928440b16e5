#!/bin/bash
# round 4: the new tests (forced variants, CLI deadline over the stand-in, bench contract, multi-GPU skips), then the two-waves-per-SIMD thresholds
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_forced_variants.py tests/test_mock_rccl.py tests/test_bench_contract.py tests/test_multi_gpu.py tests/test_rccl_transport.py tests/test_cli.py -m gpu -x -q -rs > $O/pytest_j.log 2>&1 || { tail -n 40 $O/pytest_j.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_j.log)"; grep SKIP $O/pytest_j.log | head
SHAPES="2400x2400:add 2700x2700:add 3000x3000:add 3300x3300:add 3600x3600:add 2400x2400:drain 3000x3000:drain" timeout -k 10 600 python tools/ab_shapes.py 2 "base WDPM_RELAY=0" "base WDPM_RELAY=0 WDPM_TALL_ROWS=24,12" "base WDPM_RELAY=0 WDPM_TALL_ROWS=18,9" "base WDPM_RELAY=0 WDPM_TALL_ROWS=12,6" > $O/tall_rows_sweep.txt 2>&1; cat $O/tall_rows_sweep.txt

#!/bin/bash
# round 5, session 2b: the test files session 2 did not reach (it stopped at a NameError in tests/test_mock_rccl.py), without -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s2; mkdir -p $O
timeout -k 10 1150 python -m pytest tests/test_mock_rccl.py tests/test_multi_gpu.py tests/test_rccl_transport.py tests/test_rowblock.py tests/test_seqsum_model.py tests/test_settled_golden.py tests/test_setup_stats.py tests/test_stencil_forms.py tests/test_step_floor.py -m gpu -q -rs --durations=15 > $O/pytest_gpu_b.log 2>&1 || { grep -n "Error\|^FAILED\|^E  " $O/pytest_gpu_b.log | head -40; tail -n 5 $O/pytest_gpu_b.log; exit 1; }
echo "suite (second half): $(tail -n 1 $O/pytest_gpu_b.log)"; grep -A 18 "slowest" $O/pytest_gpu_b.log | cut -c1-150

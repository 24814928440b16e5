cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/final; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -q -rs --durations=12 > $O/pytest_gpu.log 2>&1; rc=$?
echo "suite (rc $rc): $(tail -n 1 $O/pytest_gpu.log)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
exit $rc

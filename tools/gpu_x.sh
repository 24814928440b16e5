#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest.log | head -30; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 300 python bench.py > $O/bench_default.json 2>/dev/null; cut -c1-150 $O/bench_default.json
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2>/dev/null
python - <<'P'
import json
for f in ('bench_default','bench_steps20'):
    d=json.load(open(f'gpurun_out/r02/{f}.json')); r=d['roofline']
    print(f, '%.4g'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel %.4f'%r['kernel_ms_per_iteration'], 'all %.4f'%r['kernel_ms_per_iteration_all_launches'], 'frac %.3f'%r['frac'], 'frac_all %.3f'%r['frac_all_launches'], 'job %.3f'%r['job_frac'])
P

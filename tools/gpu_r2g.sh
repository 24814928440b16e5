#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r2g; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED\|assert" $O/pytest.log | head -30; exit 1; }
for tri in 1; do for sz in 482 900; do echo -n "tri=$tri add $sz: "; WDPM_TRI=$tri timeout -k 10 300 python bench.py --size $sz --steps 3000 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'us/iter', d['ms_per_step']*1e3, 'kernel', d['roofline']['kernel_ms_per_iteration']*1e3)"; done; done
for tri in 1 0; do for sz in 482 900; do echo -n "tri=$tri drain $sz: "; WDPM_TRI=$tri timeout -k 10 300 python bench.py --module drain --size $sz --steps 3000 --warmup 5 --drain-spinup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'us/iter', d['ms_per_step']*1e3)"; done; done

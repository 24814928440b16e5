#!/bin/bash
# round 3, session U: with the waves of a SIMD in step the add kernel issues VALU 95 % of its cycles - does the gate-free
# step (35 instructions of 1001 less per window step) pay now?  Parity on it, then A/B by environment (same library)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_golden.py -m gpu -x -q -k "not standin" > $O/pytest_plain_add.log 2>&1 || { tail -n 30 $O/pytest_plain_add.log; exit 1; }
echo "parity: $(tail -n 1 $O/pytest_plain_add.log)"
WDPM_PRIO=2 WDPM_RELAY=0 WDPM_TRI=0 timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dry_tiles.py -m gpu -x -q > $O/pytest_plain_add_forced.log 2>&1 || { tail -n 30 $O/pytest_plain_add_forced.log; exit 1; }
echo "parity, PRIO forced, marching only: $(tail -n 1 $O/pytest_plain_add_forced.log)"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  %.4g cell-updates/s  %.2f us/iteration  kernel %.4f ms' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_iteration']))"; }
for rep in 1 2 3; do for v in 0 1; do
  export WDPM_PLAIN_ADD=$v
  echo -n "WDPM_PLAIN_ADD=$v add 8192: "; timeout -k 10 200 python bench.py --size 8192 --steps 1000 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "WDPM_PLAIN_ADD=$v add 16384: "; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | line
  echo -n "WDPM_PLAIN_ADD=$v slab add 2 GPUs: "; timeout -k 10 200 python tools/shape_bench.py 8193 16384 100 fused add 2>/dev/null
done; done 2>&1 | tee $O/plain_add_prio_ab.txt

#!/bin/bash
# one GPU-box session: parity suite, bench, multi-rank rehearsal of bench.py on one GPU (gloo + host
# halos), config-3 chunk-height sweep at 4096^2, config-5 drain run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_round.log 2>&1 || { tail -n 30 gpurun_out/pytest_round.log; exit 1; }
tail -n 2 gpurun_out/pytest_round.log
echo "== bench default"; timeout -k 10 300 python bench.py 2>/dev/null | tee gpurun_out/bench_default.json | cut -c1-900
echo "== rehearsal: 3 ranks on one GPU, gloo + host-staged halos, 4096^2"
WDPM_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 3 --size 4096 --steps 40 --warmup 4 --no-cpu-baseline 2>&1 | grep -E '"metric"|Error|error' | cut -c1-400
echo "== same size, 1 rank"; timeout -k 10 120 python bench.py --size 4096 --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | cut -c1-300
echo "== config 3: chunk-height sweep at 4096^2 (rows per marching chunk -> cell-updates/s)"
for h in 24 48 96 192 0 384 768; do echo -n "H=$h "; WDPM_CHUNK_ROWS=$h timeout -k 10 120 python bench.py --size 4096 --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g cell-updates/s  kernel %.4f ms  frac %.3f' % (d['value'], d['roofline']['kernel_ms_per_iteration'], d['roofline']['frac']))"; done
echo "== config 5: drain, 8192^2, 1 GPU"; timeout -k 10 300 python bench.py --module drain --size 8192 --steps 1000 --warmup 5 --drain-spinup 200 --no-cpu-baseline 2>/dev/null | cut -c1-600

#!/bin/bash
# round-2 session B: self-launched ranks again, host cost of queueing for 8 rank threads, 8-slab group at 16k
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r2b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bench_contract.py tests/test_rowblock.py tests/test_rccl_transport.py tests/test_cli.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -n 3 $O/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED" $O/pytest.log | head -20; exit 1; }
echo "== bench --gpus 2 (self-launched, shared GPU -> gloo+host)"; timeout -k 10 300 python bench.py --gpus 2 --size 8192 --steps 100 --warmup 8 > $O/bench_ranks2.json 2>$O/bench_ranks2.err; cut -c1-900 $O/bench_ranks2.json; tail -n 3 $O/bench_ranks2.err
echo "== group x8 on one GPU at 2048^2 (kernels ~10 us: the host side is the limit -> cost of queueing per iteration per rank)"
timeout -k 10 300 python bench.py --gpus 8 --driver group --size 2048 --steps 2000 --warmup 40 --exchange-every 4 > $O/bench_group8_2048.json 2>$O/g.err; python - <<'P'
import json; d=json.load(open('gpurun_out/r2b/bench_group8_2048.json')); c=d['config']; print(d['ms_per_step'], c['enqueue_us_per_iteration_per_rank'], c['halo_refresh_host_us_per_iteration_per_rank'], c['decomposition'])
P
echo "== group x1 at 2048^2"; timeout -k 10 300 python bench.py --size 2048 --steps 2000 --warmup 40 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
echo "== group x8 at 16384^2, k=4 and k=8"
for k in 4 8; do timeout -k 10 300 python bench.py --gpus 8 --driver group --steps 200 --warmup 8 --exchange-every $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], c['enqueue_us_per_iteration_per_rank'], c['halo_refresh_host_us_per_iteration_per_rank'])"; done

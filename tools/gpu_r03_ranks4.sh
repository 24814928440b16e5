#!/bin/bash
# 4 rank processes sharing the one GPU, the driver's command shape at 16384^2, host-staged halos and the stand-in RCCL (IPC wire): wall time and max_diff against one rank
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
echo "== 4 ranks sharing the one GPU, the driver's command shape (wall time incl. torch import, DEM, rehearsal block); final code of round 3" | tee $O/ranks4_wall.txt
for mode in host mock; do
  if [ $mode = mock ]; then export WDPM_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so WDPM_HALO=rccl; else unset WDPM_RCCL_LIB WDPM_HALO; fi
  s=$(date +%s.%N); timeout -k 10 400 python bench.py --gpus 4 --size 16384 --steps 20 --warmup 5 > $O/ranks4_$mode.json 2> $O/ranks4_$mode.err; e=$(date +%s.%N)
  python - <<PY | tee -a $O/ranks4_wall.txt
import json
d=json.load(open("$O/ranks4_$mode.json")); c=d["config"]
print("$mode: wall %.1f s  n_gpus %d halo %s rccl_ranks %s degraded %s ms_per_step %.3f max_diff_m %r" % ($e-$s, d["n_gpus"], c["halo"], c["rccl_ranks"], d.get("degraded"), d["ms_per_step"], c["max_diff_m"]))
PY
done
unset WDPM_RCCL_LIB WDPM_HALO
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1 rank: max_diff_m %r' % d['config']['max_diff_m'])" | tee -a $O/ranks4_wall.txt

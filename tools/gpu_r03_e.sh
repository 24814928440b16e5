#!/bin/bash
# round 3, session E: the round's profile of the bench command (trace + PMC passes), the N-rank rehearsal's wall time, the
# compute side of the scaling curve at several exchange intervals
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
bash tools/profile.sh r03/bench16k > $O/bench16k_profile.txt 2>&1; grep -h '"metric"' $O/bench16k/trace.log | cut -c1-200
echo "== 4 ranks sharing the one GPU, the driver's command shape (wall time incl. torch import, DEM, rehearsal block)" | tee $O/ranks4_wall.txt
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
echo "== compute side of the scaling curve" | tee $O/scale_projection.txt
for k in 4 8 12; do echo "--- k = $k, overlapped last iteration"; timeout -k 10 300 python tools/scale_projection.py 16384 $k 40; done 2>&1 | tee -a $O/scale_projection.txt
echo "--- k = 4, plain iterations only"; timeout -k 10 300 python tools/scale_projection.py 16384 4 40 plain 2>&1 | tee -a $O/scale_projection.txt

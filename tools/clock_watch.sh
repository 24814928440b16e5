#!/bin/bash
# experiment only: shader clock, memory clock and board power as rocm-smi reads them, every ~0.1 s, while bench.py runs a long block
# (usage on the GPU box: bash tools/clock_watch.sh > gpurun_out/r04/clock_watch.txt)
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|fclk\|power" | head -8
echo "== polling while 'bench.py --steps 3000 --warmup 20' runs"
python bench.py --steps 3000 --warmup 20 --no-cpu-baseline > /tmp/bench_watch.json 2>/dev/null &
pid=$!
t0=$(date +%s%N)
while kill -0 $pid 2>/dev/null; do
  ms=$(( ($(date +%s%N) - t0) / 1000000 ))
  line=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|Graphics Package Power" | sed 's/.*: //' | tr '\n' ' ')
  echo "$ms ms  $line"
done
wait $pid
python - <<PY
import json
d = json.loads(open("/tmp/bench_watch.json").read().strip().splitlines()[-1])
print("bench: ms/step %.4f kernel %.4f job_frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms_per_iteration"], d["roofline"]["job_frac"]))
PY

#!/bin/bash
# round 4, after the last kernel-source change: the PMC passes again (profiles/traffic.json names the build), the driver's command with
# the counters quoted, warm-up sweep of that command, the command line end to end at 16384^2
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04/final; mkdir -p $O
bash tools/profile.sh r04/final/prof16k_b > $O/profile16k_b.txt 2>&1; tail -n 2 $O/profile16k_b.txt | cut -c1-100
for w in 5 30 100; do echo -n "--steps 20 --warmup $w: "; timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.4g  ms/step %.4f  kernel_ms %.4f  job_frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_per_iteration'], r['job_frac']))"; done > $O/warm_sweep.txt 2>&1; cat $O/warm_sweep.txt
bash tools/e2e_16k.sh > $O/e2e_cli_16384.txt 2>&1; cat $O/e2e_cli_16384.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
run() { timeout -k 10 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value %.4g  ms/step %.4f  kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_iteration']))"; }
{
echo "== clamp default"; run; run
echo "== WDPM_CLAMP=0"; WDPM_CLAMP=0 run; WDPM_CLAMP=0 run
echo "== r3 lib"; WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_r3_libwdpm_hip.so run
} > $O/clamp_ab2.txt 2>&1; cat $O/clamp_ab2.txt

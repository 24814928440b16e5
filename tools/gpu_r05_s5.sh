#!/bin/bash
# round 5, session 5: per-XCD step counts, us per step and the balance weights in the converged state -> gpurun_out/r05/s5/
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s5; mkdir -p $O
for sh in "4096 4096 add" "8190 1053 drain"; do
  WT_WARM=60 WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so timeout -k 10 150 python tools/wave_times.py $sh 2>&1 | grep -v amdgpu.ids
done > $O/xcd_steps.txt 2>&1
grep -E "^==|physical|balance" $O/xcd_steps.txt | cut -c1-260

#!/bin/bash
# round 5, session 11: is a slow workgroup slow again in the next launch? (per-workgroup us per step, launch against launch, and whether it sat on the same CU)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s11; mkdir -p $O
for sh in "4096 4096 add" "8190 1053 drain" "8192 8192 add" "16384 16384 add"; do WT_WARM=300 WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so timeout -k 10 150 python tools/wave_times.py $sh 2>&1 | grep -v amdgpu.ids; done > $O/workgroup_persistence.txt 2>&1
grep -E "^==|workgroups:|SIMDs" $O/workgroup_persistence.txt | cut -c1-260

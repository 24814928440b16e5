#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so
{ timeout -k 10 200 python tools/wave_times.py 16384; timeout -k 10 100 python tools/wave_times.py 16384 2116; timeout -k 10 100 python tools/wave_times.py 4096; timeout -k 10 100 python tools/wave_times.py 8190 1053 drain; } > $O/wave_times.txt 2>&1
grep -E "^==|in flight|median end by slot" $O/wave_times.txt

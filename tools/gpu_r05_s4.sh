#!/bin/bash
# round 5, session 4: what a launch boundary costs (period on the stream against the span of the waves' lifetimes), and whether a HIP
# graph of the same launches shortens it   -> gpurun_out/r05/s4/
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s4; mkdir -p $O
for sh in "8190 1053 drain" "16384 2049 add" "4096 4096 add" "8192 8192 add" "16384 16384 add"; do
  WT_WARM=200 WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so timeout -k 10 150 python tools/wave_times.py $sh 2>&1 | grep -v amdgpu.ids
done > $O/period_vs_span.txt 2>&1
grep -E "^period|^==" $O/period_vs_span.txt | cut -c1-200
for sh in "1053 8190 drain" "2049 16384 add" "4096 4096 add" "8192 8192 add"; do timeout -k 10 150 python tools/graph_probe.py $sh 2>&1 | grep -v amdgpu.ids; done > $O/graph_probe.txt 2>&1
cat $O/graph_probe.txt | cut -c1-250

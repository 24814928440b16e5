#!/bin/bash
# round 5, session 7: the first chunk row ends late (+2 .. +11 %): is it the raster's first rows or the workgroups dispatched first?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s7; mkdir -p $O
for v in wt wtrev; do for sh in "8190 1053 drain" "8192 8192 drain" "4096 4096 add"; do echo "#### $v"; WT_WARM=300 WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_${v}_libwdpm_hip.so timeout -k 10 120 python tools/wave_times.py $sh 2>&1 | grep -v amdgpu.ids; done; done > $O/first_chunk_row.txt 2>&1
grep -E "^####|^==|chunk row|last to end" $O/first_chunk_row.txt | cut -c1-330

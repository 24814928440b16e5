#!/bin/bash
# round 5, session 3: (a) the HBM ablation - what the marching kernel costs when its rows come from / go to the caches: the ceiling of
# any design that removes bytes (two iterations per pass); (b) slot filling + pairing against round 4's geometry
# (WDPM_PAIR=0) on the shapes it is for; (c) per-wave timestamps of the shapes' launches   -> gpurun_out/r05/s3/
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s3; mkdir -p $O
BENCH_ARGS="--steps 300 --warmup 20" WDPM_TILES=0 timeout -k 10 500 bash tools/ab_interleaved.sh 2 base ablate1 ablate3 ablate7 > $O/hbm_ablation.txt 2>&1 || { tail $O/hbm_ablation.txt; exit 1; }
tail -n 5 $O/hbm_ablation.txt
SHAPES="1053x8190:drain 2049x16384:add 4096x4096:add 4096x4096:drain 8192x8192:drain 8192x8192:add" timeout -k 10 420 python tools/ab_shapes.py 3 "base" "base WDPM_PAIR=0" > $O/pair_shapes_ab.txt 2>&1 || { tail $O/pair_shapes_ab.txt; exit 1; }
cat $O/pair_shapes_ab.txt
for sh in "8190 1053 drain" "16384 2049 add" "4096 4096 add"; do
  for p in 1 0; do echo "#### WDPM_PAIR=$p"; WT_WARM=200 WDPM_PAIR=$p WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_wt_libwdpm_hip.so timeout -k 10 120 python tools/wave_times.py $sh 2>&1 | grep -v amdgpu.ids; done
done > $O/wave_times_pair.txt 2>&1
grep -E "^####|^==|SIMDs holding" $O/wave_times_pair.txt | cut -c1-220

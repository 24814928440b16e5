#!/bin/bash
# one-off timing experiments around the fused kernel's HBM behaviour (see profiles/r01/README.md)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== yardstick"; timeout -k 10 200 tools/_build/hbm_yardstick
echo "== base vs no-compute ablation, 16384^2"; bash tools/ab.sh nocomp
echo "== shapes (fused)"
for s in "16384 16384" "131072 2050" "1572864 171" "524288 512"; do timeout -k 10 200 python tools/shape_bench.py $s 60 fused; done
echo "== shapes, no-compute"
export WDPM_HIP_LIB=$PWD/wdpm_amd/csrc/alt_nocomp_libwdpm_hip.so
for s in "16384 16384" "1572864 171"; do timeout -k 10 200 python tools/shape_bench.py $s 60 fused; done

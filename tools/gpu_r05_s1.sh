#!/bin/bash
# round 5, session 1: the whole -m gpu suite on the new code, then the HBM ablation (what the marching kernel costs when its rows
# come from / go to the caches: the ceiling of any design that removes bytes) -> gpurun_out/r05/s1/
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/s1; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > $O/pytest_gpu.log 2>&1 || { tail -n 40 $O/pytest_gpu.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest_gpu.log)"
BENCH_ARGS="--steps 300 --warmup 20" WDPM_TILES=0 timeout -k 10 600 bash tools/ab_interleaved.sh 2 base ablate1 ablate3 ablate7 > $O/hbm_ablation.txt 2>&1
tail -n 6 $O/hbm_ablation.txt

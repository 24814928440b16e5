#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02
echo "--- plain iterations only (WDPM_OVERLAP=0)" | tee -a gpurun_out/r02/scale_projection.txt
python tools/scale_projection.py 16384 4 50 plain 2>/dev/null | cut -c1-150 | tee -a gpurun_out/r02/scale_projection.txt

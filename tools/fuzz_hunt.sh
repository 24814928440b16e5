#!/bin/bash
# A long hunt with the randomised GPU tests (seeds far outside the ranges the suite runs): usage on the GPU box: bash tools/fuzz_hunt.sh <lo> <hi>
cd $GRAFT_REPO_ROOT; lo=${1:-10000}; hi=${2:-14000}; mkdir -p gpurun_out
for t in "tests/test_hip_parity.py -k random_call" "tests/test_dry_tiles.py -k random_ponds" "tests/test_rowblock.py -k hip_random_group" "tests/test_setup_stats.py -k random_shapes_hip" "tests/test_hip_parity.py -k adversarial"; do
  echo "== $t seeds $lo:$hi"; WDPM_FUZZ_SEEDS=$lo:$hi timeout -k 10 900 python -m pytest $t -m gpu -q -x 2>&1 | tail -n 6
done
echo "== CLI differential seeds $lo:$((lo + 400))"; WDPM_FUZZ_SEEDS=$lo:$((lo + 400)) timeout -k 10 900 python -m pytest tests/test_cli_differential.py -m gpu -q -x 2>&1 | tail -n 6
echo "== the same call sequences and row-block jobs with six rows per triangle wave forced (WDPM_TRI_K=2)"
for t in "tests/test_hip_parity.py -k random_call" "tests/test_rowblock.py -k hip_random_group" "tests/test_hip_parity.py -k every_"; do
  WDPM_TRI_K=2 WDPM_FUZZ_SEEDS=$lo:$hi timeout -k 10 900 python -m pytest $t -m gpu -q -x 2>&1 | tail -n 3
done
echo "== the same with the relay kernel's workgroups at eight waves forced everywhere (WDPM_RELAY=2 WDPM_RELAY_NW=8), and with the relay kernel off"
for t in "tests/test_hip_parity.py -k random_call" "tests/test_rowblock.py -k hip_random_group" "tests/test_hip_parity.py -k every_"; do
  WDPM_RELAY=2 WDPM_RELAY_NW=8 WDPM_FUZZ_SEEDS=$lo:$hi timeout -k 10 900 python -m pytest $t -m gpu -q -x 2>&1 | tail -n 3
  WDPM_RELAY=0 WDPM_FUZZ_SEEDS=$lo:$hi timeout -k 10 900 python -m pytest $t -m gpu -q -x 2>&1 | tail -n 3
done
echo "== the same on the marching kernel alone (WDPM_RELAY=0 WDPM_TRI=0): chunk heights from skewed per-XCD weights and the DEM codes on every launch (WDPM_BALANCE=2 WDPM_DEM32=2), then gated / unclamped / no priorities / no 16-bit offsets"
for t in "tests/test_hip_parity.py -k random_call" "tests/test_rowblock.py -k hip_random_group" "tests/test_dry_tiles.py -k random_ponds" "tests/test_hip_parity.py -k every_"; do
  case "$t" in *dry_tiles*) b="";; *) b="WDPM_BALANCE=2";; esac      # (the forced table drops the dry-tile flags those tests look at)
  env $b WDPM_DEM32=2 WDPM_RELAY=0 WDPM_TRI=0 WDPM_FUZZ_SEEDS=$lo:$hi timeout -k 10 900 python -m pytest $t -m gpu -q -x 2>&1 | tail -n 3
  WDPM_PLAIN=0 WDPM_CLAMP=0 WDPM_PRIO=0 WDPM_DEM16=0 WDPM_RELAY=0 WDPM_TRI=0 WDPM_FUZZ_SEEDS=$lo:$hi timeout -k 10 900 python -m pytest $t -m gpu -q -x 2>&1 | tail -n 3
done
echo "== (round 5) the marching kernel alone with every slot of the resident round filled and partner strips rounded half a triple apart (WDPM_BALANCE=2 WDPM_PAIR=2)"
for t in "tests/test_hip_parity.py -k random_call" "tests/test_rowblock.py -k hip_random_group" "tests/test_hip_parity.py -k every_"; do
  WDPM_BALANCE=2 WDPM_PAIR=2 WDPM_RELAY=0 WDPM_TRI=0 WDPM_FUZZ_SEEDS=$lo:$hi timeout -k 10 900 python -m pytest $t -m gpu -q -x 2>&1 | tail -n 3
done

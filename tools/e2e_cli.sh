#!/bin/bash
# end-to-end wall time of the WDPMCL drop-in on a synthetic n x n DEM, by phase (WDPM_TIMING=1):
#   tools/e2e_cli.sh [n]      (needs tools/_build/synth_asc: see tools/synth_asc.c)
cd $GRAFT_REPO_ROOT; N=${1:-8192}; W=/tmp/e2e_$N; mkdir -p $W
BIN=$PWD/wdpm_amd/bin/WDPMCL
SECONDS=0; tools/_build/synth_asc $N $W/dem.asc; echo "synth_asc ${SECONDS} s"
ls -la $W/dem.asc | awk '{print "dem.asc bytes", $5}'
cd $W
run() { echo "--- $*"; ( "$@" | grep -E "^ +[0-9]+ |Run Time|Final volume" | tail -n 5 ) 2>&1 | grep -v amdgpu.ids; }
export WDPM_TIMING=1
echo "== add 100 mm, 2000 iterations, pinned staging"; WDPM_PINNED=1 run $BIN add dem.asc NULL out.asc NULL 100 1.0 1.0 1 1 0.005 2000
echo "== same, pageable staging";                     WDPM_PINNED=0 run $BIN add dem.asc NULL out0.asc NULL 100 1.0 1.0 1 1 0.005 2000
cmp out.asc out0.asc && echo "outputs identical"
echo "== add with checkpoint (text + f64 sidecar), 3000 iterations"; WDPM_SCRATCH_BINARY=1 run $BIN add dem.asc NULL out2.asc scr.asc 100 1.0 1.0 1 1 0.005 3000
echo "== drain from out.asc, 3000 iterations"; run $BIN drain dem.asc out.asc outd.asc NULL 0.1 1.0 1 1 0.005 3000
rm -rf $W

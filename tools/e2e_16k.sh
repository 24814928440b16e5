cd $GRAFT_REPO_ROOT; N=16384; W=/tmp/e2e_$N; mkdir -p $W
df -h /tmp | tail -1
SECONDS=0; tools/_build/synth_asc $N $W/dem.asc; echo "synth_asc ${SECONDS} s"; ls -la $W/dem.asc | awk '{print "dem.asc bytes", $5}'
cd $W; export WDPM_TIMING=1
$GRAFT_REPO_ROOT/wdpm_amd/bin/WDPMCL add dem.asc NULL out.asc NULL 100 1.0 1.0 1 1 0.005 1000 2>&1 | grep -E "timing|Run Time|^ +1000 " | grep -v amdgpu
rm -rf $W

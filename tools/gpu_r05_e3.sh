#!/bin/bash
# round 5, evidence part 3: smoke(), the warm-up sweep of the driver's command, WDPMCL end to end at 16384^2, a fuzz hunt on the final code
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05/final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
bash tools/evidence_session.sh r05 sweep e2e
timeout -k 10 900 bash tools/fuzz_hunt.sh 110000 110600 > $O/fuzz_hunt5.txt 2>&1; grep -E "passed|failed|error" $O/fuzz_hunt5.txt | sort | uniq -c | sort -rn | head -12

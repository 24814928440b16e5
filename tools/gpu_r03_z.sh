#!/bin/bash
# round 3, last session: the whole -m gpu suite on the final code, then a hunt with the randomised tests on seeds outside the suite's
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -n 30 $O/pytest.log; exit 1; }
echo "suite: $(tail -n 1 $O/pytest.log)"
bash tools/fuzz_hunt.sh 30000 31200 > $O/fuzz_hunt.txt 2>&1; grep -E "passed|failed|error" $O/fuzz_hunt.txt | sort | uniq -c

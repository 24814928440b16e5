cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05/final
timeout -k 10 1100 bash tools/fuzz_hunt.sh 130000 130900 > gpurun_out/r05/final/fuzz_hunt7.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r05/final/fuzz_hunt7.txt | sort | uniq -c | sort -rn | head -8; grep -ci "failed" gpurun_out/r05/final/fuzz_hunt7.txt

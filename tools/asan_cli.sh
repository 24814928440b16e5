#!/bin/bash
# AddressSanitizer + UBSan, then ThreadSanitizer, runs of the WDPMCL host code (CPU build against the oracle back-end; GPU ASan
# is not available on the pool): plain, threaded I/O, three slabs (rank threads, peer-copy halos, barrier / all-gather between
# them), checkpoint sidecar + resume, drain (one and three slabs), subtract, an "oversized" raster cut into slabs, 300 random jobs.  Prints any sanitizer report; silence = clean.   usage: bash tools/asan_cli.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd); W=$(mktemp -d); trap 'rm -rf $W' EXIT
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -o $W/WDPMCL_asan \
    $R/wdpm_amd/csrc/wdpmcl_main.c $R/wdpm_amd/csrc/arcascii.c $R/oracle/wdpm_oracle.c $R/wdpm_amd/csrc/synth.c \
    $R/wdpm_amd/csrc/wdpm_rowblock.c -lpthread -lm
cd $W; zcat $R/tests/golden/basin5.asc.gz > basin5.asc
run() { echo "== $*"; env "$@" > out.txt 2> err.txt || echo "exit code $?"; grep -E "ERROR|runtime error|leak" err.txt || true; }
run ./WDPMCL_asan add basin5.asc NULL a.asc s.asc 100 1.0 1.0 0 0 0.005 2000
run WDPM_HOST_PAR_MIN=1 WDPM_IO_THREADS=6 ./WDPMCL_asan add basin5.asc NULL a.asc s.asc 100 1.0 1.0 0 0 0.005 2000
run WDPM_DEVICES=0,0,0 WDPM_EXCHANGE_EVERY=2 ./WDPMCL_asan add basin5.asc NULL a.asc s.asc 100 1.0 1.0 0 0 0.005 2000
run WDPM_SCRATCH_BINARY=1 ./WDPMCL_asan add basin5.asc NULL a.asc s.asc 100 1.0 1.0 0 0 0.005 2000
run WDPM_SCRATCH_BINARY=1 ./WDPMCL_asan add basin5.asc NULL a2.asc s.asc 100 1.0 1.0 0 0 0.005 1000
run ./WDPMCL_asan drain basin5.asc a.asc d.asc NULL 1.0 1.0 0 0 0.005 1000
run WDPM_DEVICES=0,0,0 WDPM_EXCHANGE_EVERY=3 ./WDPMCL_asan drain basin5.asc a.asc d3.asc NULL 1.0 1.0 0 0 0.005 1000
run ./WDPMCL_asan subtract basin5.asc a.asc sub.asc NULL 10 1.0 0 0 0.005 1000
run WDPM_MAX_SLAB_CELLS=60000 ./WDPMCL_asan add basin5.asc NULL a4.asc NULL 100 1.0 1.0 0 0 0.005 1000
# the random jobs of tests/cli_fuzz.py (tiny rasters, odd formatting, parameter files, resume, 2-5 row blocks) through the sanitizer
# build, against the reference executable: a sanitizer report ends the job with a non-zero status, i.e. shows up as a mismatch
if [ -x $R/oracle/_ref/WDPMCL_ref ]; then
  echo "== 300 random jobs"; python3 - $R $W <<'PY'
import sys; sys.path.insert(0, sys.argv[1] + "/tests")
from cli_fuzz import one
bad = sum(not one(seed, sys.argv[2] + "/fuzz", sys.argv[1] + "/oracle/_ref/WDPMCL_ref", sys.argv[2] + "/WDPMCL_asan", seed % 2 == 1)[0] for seed in range(300))
print("mismatches / sanitizer reports:", bad)
PY
fi
# ThreadSanitizer on the threaded parts (checkpoint writer thread, rank threads of the row-block driver, threaded ArcASCII I/O)
gcc -O1 -g -fsanitize=thread -ffp-contract=off -o $W/WDPMCL_tsan \
    $R/wdpm_amd/csrc/wdpmcl_main.c $R/wdpm_amd/csrc/arcascii.c $R/oracle/wdpm_oracle.c $R/wdpm_amd/csrc/synth.c \
    $R/wdpm_amd/csrc/wdpm_rowblock.c -lpthread -lm
echo "== tsan"; WDPM_IO_THREADS=6 WDPM_SCRATCH_BINARY=1 ./WDPMCL_tsan add basin5.asc NULL a.asc s.asc 100 1.0 1.0 0 0 0.005 3000 > out.txt 2> err.txt || echo "exit code $?"
grep -c "WARNING: ThreadSanitizer" err.txt || true
echo "== tsan, four slabs"; WDPM_DEVICES=0,0,0,0 WDPM_EXCHANGE_EVERY=2 WDPM_IO_THREADS=4 ./WDPMCL_tsan drain basin5.asc a.asc d4.asc s2.asc 1.0 1.0 0 0 0.005 2000 > out.txt 2> err.txt || echo "exit code $?"
grep -c "WARNING: ThreadSanitizer" err.txt || true
echo done

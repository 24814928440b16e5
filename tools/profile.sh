#!/bin/bash
# Round profile: kernel trace + stats, then the two HBM-counter passes, each its own rocprofv3 run.
# usage (on the GPU box): bash tools/profile.sh <name> [extra bench.py arguments]    -> gpurun_out/<name>/
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift; X="$*"
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline $X > $OUT/trace.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline $X > $OUT/fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline $X > $OUT/write.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline $X > $OUT/sq.log 2>&1
cd $R
grep -h '"metric"' $OUT/trace.log | cut -c1-400
cat $OUT/trace/*/*_kernel_stats.csv
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]; res = {}
for d in ('fetch', 'write', 'sq'):
    for f in glob.glob(f'{out}/{d}/*/*_counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        seen = set()
        for r in csv.DictReader(open(f)):
            # template arguments kept: <0,..> is add/subtract, <2,..> drain, the third the DEM-code variant
            name = r['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0].replace('void ', '').strip()
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
            if d == 'sq' and r['Dispatch_Id'] not in seen:      # the kernel's duration while the counters were taken
                seen.add(r['Dispatch_Id'])
                agg[name]['kernel_ms'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
        for k, c in agg.items():
            for cn, v in c.items():
                res.setdefault(k, {})[cn] = {"n": len(v), "mean": sum(v) / len(v)}
json.dump(res, open(out + '/pmc_summary.json', 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if 'fused' in k or k in ('flush_snapshot_kernel', 'max_diff_kernel')}, indent=1))
PY

#!/bin/bash
# rocprofv3 PMC passes (one counter group per run, as the MI355X guide prescribes) on a short bench run
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/g$i.log 2>&1 || { echo "group $i failed: $grp"; tail -n 5 $OUT/g$i.log; }
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + '/g*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(f)):
        if 'fused' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    if dur: print('# kernel ms %.4f' % (sum(dur) / len(dur)))
    for k, v in agg.items(): print('%-40s %.5g' % (k, sum(v) / len(v)))
PY

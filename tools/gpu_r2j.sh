#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r2j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/sparse -- python3 $GRAFT_REPO_ROOT/tools/sparse_bench.py 16384 100 2 > $GRAFT_REPO_ROOT/$O/sparse.log 2>&1
cd $GRAFT_REPO_ROOT; tail -n 4 $O/sparse.log; find $O/sparse -name "*kernel_stats.csv" | head -1 | xargs -r head -6 | cut -c1-220
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/r2j/sparse/*/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'fused_iteration' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print('fused launches', len(d), 'first 8 us', [round(x) for x in d[:8]], 'median of sparse-mode part', sorted(d[110:205])[47] if len(d)>205 else None)
gaps=[(int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']))/1e3 for i in range(120,200)]
print('gaps between launches us', round(sum(gaps)/len(gaps),1), 'grid', rows[150]['Grid_Size_X'] if 'Grid_Size_X' in rows[150] else rows[150].get('Grid_Size'))
P

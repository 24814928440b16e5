#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r2l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do n=$(echo $c | cut -d' ' -f1); timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/$n -- python3 $GRAFT_REPO_ROOT/tools/sparse_bench.py 16384 20 1 > $GRAFT_REPO_ROOT/$O/$n.log 2>&1; done
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv,glob,collections
for d in ('WRITE_SIZE','FETCH_SIZE','SQ_WAVES'):
    f=glob.glob(f'gpurun_out/r2l/{d}/*/*_counter_collection.csv')[0]
    rows=[r for r in csv.DictReader(open(f)) if 'fused_iteration' in r['Kernel_Name'] and 'true, false>' in r['Kernel_Name']]
    by=collections.defaultdict(list)
    for r in rows: by[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in by.items(): print(d,k,'n',len(v),'first',v[:8],'last',v[-4:])
P

#!/bin/bash
# SQ issue/wait breakdown of the Lloyd assign kernel (unpruned path)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ET_KMEANS_PRUNE=${ET_KMEANS_PRUNE:-0}
rm -rf /tmp/sq && timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d /tmp/sq -- python $R/tools/kbench.py 1e7 km 12 > /tmp/sq.log 2>&1
tail -3 /tmp/sq.log
python - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/sq/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][-60:]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'assign' not in k: continue
    print(k)
    for c,vals in v.items():
        vals=vals[2:] if len(vals)>4 else vals
        print(f"   {c:24s} n={len(vals):3d} mean={sum(vals)/len(vals):.4g}")
PY

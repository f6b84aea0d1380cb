#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
rm -rf /tmp/sq && timeout 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/sq -- python $R/tools/kbench.py 1e7 desc > /tmp/sq.log 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/sq/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][-60:]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'gram_tile' not in k: continue
    for c,vals in v.items():
        print(f"   {c:28s} n={len(vals):3d} mean={sum(vals)/len(vals):.4g}")
PY
done

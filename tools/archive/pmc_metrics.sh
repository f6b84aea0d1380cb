#!/bin/bash
# SQ counters of the fused metrics kernels (ON THE GPU BOX): instructions by class and wait breakdown per launch.
#     tools/pmc_metrics.sh   -> gpurun_out/pmc_metrics.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_metrics.txt; mkdir -p $R/gpurun_out; : > $OUT
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"; do
  D=/tmp/pmc_$RANDOM; rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $D -- python $R/tools/ab_metrics.py 2000000 > /dev/null 2>&1
  python - "$D" <<'P' >> $OUT
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "metrics" not in k: continue
        k = k.split("(")[0][-45:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
P
done
cat $OUT

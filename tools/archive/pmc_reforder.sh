#!/bin/bash
# SQ / TCC counters of the reference-order Lloyd kernels at N = 1e7 (ON THE GPU BOX): instructions by class, wait breakdown,
# LDS conflicts, bytes fetched / written per launch.      tools/pmc_reforder.sh [N]  -> gpurun_out/pmc_reforder.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-10000000}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_reforder.txt; mkdir -p $R/gpurun_out; : > $OUT
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
            "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_WAIT_INST_ANY SQ_INSTS_FLAT"; do
  D=/tmp/pmc_$RANDOM; rm -rf $D
  PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $D -- python $R/tools/time_reforder.py $N > $D.log 2>&1 || tail -3 $D.log
  python - "$D" <<'P' >> $OUT
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "reforder_groups" not in k and "reforder_update_kernel2" not in k: continue
        k = k.split("(")[0][-48:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
P
done
cat $OUT

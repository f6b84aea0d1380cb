#!/bin/bash
# Profiles of one round, to be run ON THE GPU BOX through gpurun:
#     gpurun --timeout 1500 -- 'bash tools/profile_round.sh r02a'
# Writes gpurun_out/prof_<tag>/{stats,fetch,write,sq1,sq2,sq3}/ (rocprofv3 csv) and the text summaries
# gpurun_out/prof_<tag>/<tag>_*.txt|json, which are what gets copied into profiles/ (tracked).
# Counter passes are separate runs with --kernel-trace only (never combined with other trace domains).
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"

# 0. the plain bench line (default command) next to the profiles, the box's bandwidth ceilings, and the sharded path
#    forced onto one GPU (RCCL world 1): native entry points vs the torch.distributed step API
python $R/bench.py 2> "$OUT/bench.err" | tail -1 > "$OUT/${TAG}_bench_line.json"
python $R/tools/bw_ceiling.py > "$OUT/${TAG}_bw_ceiling.json" 2>/dev/null
for mode in native torch; do
    ET_BENCH_FORCE_DIST=1 ET_BENCH_DIST=$mode python $R/bench.py --no-cpu-baseline --no-extras --steps 5 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_forced_dist_$mode.json"
done

# 1. kernel trace + stats of the headline step only (N = 1e7; no extra stages, no CPU leg): the per-kernel averages
#    here are what bench.py's roofline.avg_launch_ms must agree with
rm -rf "$OUT/stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
    python $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"

# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
for C in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/$(echo $C | tr A-Z a-z | sed 's/_size//')
    rm -rf "$d"
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$d" -- $BENCH > "$d.log" 2>&1
done

# 3. SQ breakdown (issue / busy / wait), three passes of <= 8 counters
i=0
for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU"; do
    i=$((i + 1))
    rm -rf "$OUT/sq$i"
    timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/sq$i" -- $BENCH > "$OUT/sq$i.log" 2>&1
done
python $R/tools/profile_summary.py "$OUT" "$TAG"

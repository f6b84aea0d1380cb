#!/bin/bash
# Average duration of the Lloyd kernels of the bench step under rocprofv3 (ON THE GPU BOX), for same-box comparisons of
# library variants / environment switches where the event-based lloyd_us/iter is too coarse (it averages runs of four launches out of eight; until round 5: every 8th
# launch INCLUDING the exact first iteration):
#     tools/kstat.sh <label> [VAR=value ...]
LABEL=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=/tmp/kstat_$LABEL
rm -rf "$D"; cd /tmp && export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- \
    python $R/bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 > "$D.json" 2> "$D.err"
python - "$D" "$LABEL" <<'P'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
for r in rows:
    if "lloyd" in r["Name"] or "pack_kernel" in r["Name"]:
        print(f'{sys.argv[2]:>14s}  {r["Name"][:60]:60s} calls {r["Calls"]:>5s}  avg_us {float(r["AverageNs"]) / 1e3:8.2f}')
P

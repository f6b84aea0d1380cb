#!/bin/bash
# same-box sweep: threads per workgroup of the chained Lloyd kernel (and the persistent loop) over small shard sizes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for n in ${SIZES:-2e4 7e4 1e5 2e5}; do
 for cfg in ${CFGS:-"persist 768" "chain 256" "chain 512" "chain 768"}; do
  set -- $cfg
  line=$(ET_OPT_KMEANS_LOOP=$1 ET_OPT_KMEANS_FILTER_THREADS=$2 timeout 300 python $R/bench.py --trajectories $n --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
  echo "n=$n loop=$1 threads=$2 $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); st=j["stages"]; print("step_ms", j["ms_per_step"], "lloyd_ms", st["kmeans_lloyd"]["ms"], "its", st["kmeans_lloyd"]["iterations"])')"
 done
done

#!/bin/bash
# same-box A/B over shard sizes: Lloyd iterations on the packed copy against the fp32 filter (where should the packed path start?)
R=$(cd "$(dirname "$0")/.." && pwd)
for n in ${SIZES:-3e5 5e5 1e6 2e6 4e6}; do
  for rep in 1 2; do
    for v in 1 0; do
      line=$(ET_OPT_KMEANS_PACKED=$v timeout 300 python $R/bench.py --trajectories $n --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
      echo "n=$n packed=$v $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); st=j["stages"]; print("step_ms", j["ms_per_step"], "lloyd_ms", st["kmeans_lloyd"]["ms"], "its", st["kmeans_lloyd"]["iterations"])')"
    done
  done
done

#!/bin/bash
# the bench step at the reference's own shard sizes (default library policy), ms per step and per 100 Lloyd iterations
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for n in ${SIZES:-2e4 4e4 7e4 1e5 2e5 3e5 1e6}; do
  line=$(timeout 300 python $R/bench.py --trajectories $n --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
  echo "n=$n $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); st=j["stages"]; print("step_ms", j["ms_per_step"], "lloyd_ms", st["kmeans_lloyd"]["ms"], "init", st["kmeans_init"]["ms"], "fit", st["fit"]["ms"])')"
done

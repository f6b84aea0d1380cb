#!/bin/bash
# same-box A/B of the bench step over values of one environment switch:  tools/ab_env.sh VAR v1 v2 ...
R=$(cd "$(dirname "$0")/.." && pwd)
VAR=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    line=$(env $VAR=$v timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
    echo "$VAR=$v $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; st=j["stages"]; print("step_ms", j["ms_per_step"], "lloyd_ms", st["kmeans_lloyd"]["ms"], "its", st["kmeans_lloyd"]["iterations"], "fit", st["fit"]["ms"], "init", st["kmeans_init"]["ms"])')"
  done
done

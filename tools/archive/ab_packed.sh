#!/bin/bash
# same-box A/B of the bench step: Lloyd iterations on the packed copy (default) against the fp32 filter (ET_OPT_KMEANS_PACKED=0)
R=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2 3; do
  for v in 1 0; do
    line=$(ET_OPT_KMEANS_PACKED=$v timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
    echo "packed=$v $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; st=j["stages"]; print("step_ms", j["ms_per_step"], "lloyd_us/iter", round(1e3*r["avg_launch_ms"]/r["lloyd_iterations_per_launch"],2), "lloyd_ms", st["kmeans_lloyd"]["ms"], "its", st["kmeans_lloyd"]["iterations"], "fit", st["fit"]["ms"], "init", st["kmeans_init"]["ms"])')"
  done
done

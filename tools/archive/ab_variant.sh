#!/bin/bash
# same-box A/B of the bench step between the default library and variants built with tools/build_variant.sh:
#     tools/ab_variant.sh <variant> [<variant> ...]        (alternates default / variant three times)
R=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2 3; do
  for v in "" "$@"; do
    lib=$R/eigentrajectory_amd/libetamd.so
    [ -n "$v" ] && lib=$R/eigentrajectory_amd/variants/libetamd_$v.so
    line=$(ET_LIBETAMD=$lib timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
    echo "${v:-default} $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; st=j["stages"]; print("step_ms", j["ms_per_step"], "lloyd_us/iter", round(1e3*r["avg_launch_ms"]/r["lloyd_iterations_per_launch"],2), "lloyd_ms", st["kmeans_lloyd"]["ms"], "fit", st["fit"]["ms"], "proj", st["project"]["ms"], "rec", st["reconstruct"]["ms"], "init", st["kmeans_init"]["ms"])')"
  done
done

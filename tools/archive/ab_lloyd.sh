#!/bin/bash
# Where does a Lloyd iteration's time go?  Same-box A/B of the chained kernel (one launch per iteration, N = 1e7) built
# with parts of the assignment removed (results are wrong in the variants; only the launch time is looked at):
#   full | noload (no global loads) | nomfma (no MFMA, no top-2) | noload+nomfma
# usage (on the GPU box, after tools/build_variant.sh for each variant): tools/ab_lloyd.sh
R=$(cd "$(dirname "$0")/.." && pwd)
for v in "" noload nomfma noboth; do
  lib=$R/eigentrajectory_amd/libetamd.so
  [ -n "$v" ] && lib=$R/eigentrajectory_amd/variants/libetamd_$v.so
  [ -f "$lib" ] || continue
  for loop in chain persist; do
    line=$(ET_LIBETAMD=$lib ET_OPT_KMEANS_LOOP=$loop timeout 300 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
    echo "${v:-full} $loop $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("launch_ms", r["avg_launch_ms"], "iters/launch", r["lloyd_iterations_per_launch"], "us/iter", round(1e3*r["avg_launch_ms"]/r["lloyd_iterations_per_launch"],2), "lloyd_ms", j["stages"]["kmeans_lloyd"]["ms"], "its", j["stages"]["kmeans_lloyd"]["iterations"])')"
  done
done

#!/bin/bash
# same-box comparison of library variants (tools/build_variant.sh) by rocprofv3 kernel averages: tools/kab.sh <variant> ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
for v in "" "$@"; do
  lib=$R/eigentrajectory_amd/libetamd.so; [ -n "$v" ] && lib=$R/eigentrajectory_amd/variants/libetamd_$v.so
  bash $R/tools/kstat.sh "${v:-default}" ET_LIBETAMD=$lib | grep lloyd
done; done

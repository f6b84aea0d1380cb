#!/bin/bash
# PC sampling of the Lloyd kernel of the bench step (ON THE GPU BOX): where do its wavefronts sit?
#   tools/pcsample_lloyd.sh [stochastic|host_trap] -> gpurun_out/pcsample_<method>/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
M=${1:-host_trap}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/pcsample_$M; rm -rf $D; mkdir -p $D
UNIT=time; INT=1
[ "$M" = stochastic ] && UNIT=cycles && INT=1048576
ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $M \
    --pc-sampling-interval $INT --kernel-trace --output-format csv -d $D/raw -- \
    python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $D/bench.json 2> $D/err.txt
echo "rc=$?"; tail -5 $D/err.txt
find $D/raw -name "*.csv" | head; for f in $(find $D/raw -name "*pc_sampling*.csv" | head -2); do head -5 $f; wc -l $f; done

#!/bin/bash
# Build a kernel-variant copy of libetamd.so for same-box A/B runs:
#     tools/build_variant.sh <name> <source.hip> "<extra hipcc flags>"
# -> eigentrajectory_amd/variants/libetamd_<name>.so (git-ignored, travels with gpurun); select it with
#    ET_LIBETAMD=$PWD/eigentrajectory_amd/variants/libetamd_<name>.so
set -e
NAME=$1; SRC=$2; FLAGS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/eigentrajectory_amd/csrc
V=$R/eigentrajectory_amd/variants
mkdir -p "$V"
make -C "$C" -s
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
EXTRA=""
[ "$SRC" = "et_kmeans.hip" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
[ "$SRC" = "et_descriptor.hip" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
[ "$SRC" = "et_fit.hip" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
[ "$SRC" = "et_kmeans_reforder.hip" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
/opt/rocm/bin/hipcc $BASE $EXTRA $FLAGS -c "$C/$SRC" -o "$V/${SRC%.hip}_$NAME.o"
OBJS=""
for f in et_abi et_options et_trajnorm et_descriptor et_train et_fit et_kmeans et_kmeanspp et_kmeans_reforder et_sharded; do
    if [ "$f.hip" = "$SRC" ]; then OBJS="$OBJS $V/${f}_$NAME.o"; else OBJS="$OBJS $C/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$V/libetamd_$NAME.so" $OBJS -ldl
echo "$V/libetamd_$NAME.so"

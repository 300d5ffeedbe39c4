#!/bin/bash
# One translation unit rebuilt with extra flags and linked with the other objects of the in-place build into
# flux3d.jl_amd/lib/libflux3d_hip_<name>.so (for tools/ab_two_libs.sh / FX3D_HIP_LIB):
#   bash tools/build_variant.sh <name> <unit, e.g. chamfer_bwd> "<extra hipcc flags>"
set -e
NAME=$1; UNIT=$2; EXTRA=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/flux3d.jl_amd/csrc
make -j8 >/dev/null
mkdir -p /tmp/fxv_$NAME
SRC=$UNIT.hip; [ -f $SRC ] || SRC=$UNIT.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -munsafe-fp-atomics \
  -mllvm -amdgpu-mfma-vgpr-form -I../../include -Wno-unused-function $EXTRA -x hip -c $SRC -o /tmp/fxv_$NAME/$UNIT.o -save-temps=obj 2>/dev/null
OBJS=$(ls ../lib/obj/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libflux3d_hip_$NAME.so $OBJS /tmp/fxv_$NAME/$UNIT.o -ldl
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|name):" /tmp/fxv_$NAME/*gfx950.s | paste - - - | grep -v reduce_partials | sed "s/^/[$NAME] /"

#!/bin/bash
# A/B builds of ONE translation unit with extra -D flags, linked against the other objects of the shipped library:
#   tools/build_ab.sh <name> <file.hip> "<flags>"   ->  balm_amd/lib/ab/libbalm_hip_<name>.so   (BALM_HIP_LIB=<that> python ...)
set -e
NAME=$1; SRC=$2; FLAGS=$3
cd "$(dirname "$0")/.."
EXTRA=""; [ "$SRC" = "kernels_voxel.hip" ] && EXTRA="-ffp-contract=off"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wall -Wno-unused-value -Wno-unused-result -Wno-unused-function $EXTRA $FLAGS \
  -c balm_amd/csrc/$SRC -o balm_amd/lib/ab/${NAME}.o
OBJS=""
for o in kernels_accum kernels_solve kernels_build kernels_voxel kernels_cov kernels_syrk_i8 balm_multi balm_capi; do
  if [ "$o.hip" = "$SRC" ]; then OBJS="$OBJS balm_amd/lib/ab/${NAME}.o"; else OBJS="$OBJS balm_amd/lib/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o balm_amd/lib/ab/libbalm_hip_${NAME}.so $OBJS -ldl -pthread
echo balm_amd/lib/ab/libbalm_hip_${NAME}.so

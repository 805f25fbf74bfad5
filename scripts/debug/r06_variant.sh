#!/bin/bash
# Builds a variant of the library for A/B runs (PBSO_LIB=...):   scripts/debug/r06_variant.sh NAME "-DFLAG ..." [all]
#   default: only kernels_block.hip is recompiled (the R = 4 builds: -DPBSO_ONLY_R4, 40 s) and linked with the product's other objects;
#   "all": every source is recompiled with the flags (a flag that changes kernels.h).
# Output: openpbso_amd/variants/lib_NAME.so (git-ignored; travels to the GPU box).
set -e
NAME=$1; FLAGS=$2; MODE=${3:-block}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/openpbso_amd/csrc
OUT=$ROOT/build/variants/$NAME
mkdir -p $OUT $ROOT/openpbso_amd/variants
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I$ROOT/include -I$SRC $FLAGS"
hipcc $COMMON -fno-slp-vectorize -DPBSO_ONLY_R4 -c $SRC/kernels_block.hip -o $OUT/kernels_block.o 2> $OUT/kernels_block.log &
if [ "$MODE" = all ]; then
  hipcc $COMMON -fno-slp-vectorize -c $SRC/kernels_iir.hip -o $OUT/kernels_iir.o &
  hipcc $COMMON -fno-slp-vectorize -c $SRC/kernels_scan.hip -o $OUT/kernels_scan.o &
  hipcc $COMMON -fno-slp-vectorize -c $SRC/kernels_pipe.hip -o $OUT/kernels_pipe.o &
  hipcc $COMMON -ffp-contract=off -c $SRC/kernels_exact.hip -o $OUT/kernels_exact.o &
  for f in engine loaders capi group; do hipcc $COMMON -c $SRC/$f.cpp -o $OUT/$f.o & done
  wait
  OBJS="$OUT/kernels_iir.o $OUT/kernels_block.o $OUT/kernels_scan.o $OUT/kernels_pipe.o $OUT/kernels_exact.o $OUT/engine.o $OUT/loaders.o $OUT/capi.o $OUT/group.o"
else
  wait
  OBJS="$SRC/kernels_iir.o $OUT/kernels_block.o $SRC/kernels_scan.o $SRC/kernels_pipe.o $SRC/kernels_exact.o $SRC/engine.o $SRC/loaders.o $SRC/capi.o $SRC/group.o"
fi
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o $ROOT/openpbso_amd/variants/lib_$NAME.so
echo "built openpbso_amd/variants/lib_$NAME.so"

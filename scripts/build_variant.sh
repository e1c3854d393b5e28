#!/bin/bash
# build pea_diffusion_amd/libpea_hip_<tag>.so: the in-tree objects with ONE translation unit recompiled under extra flags
# usage: scripts/build_variant.sh <tag> <unit: attention|gemm|norm|...> "<extra flags>"
set -e
TAG=$1; UNIT=$2; EXTRA=$3
cd "$(dirname "$0")/../pea_diffusion_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-inline-asm -mllvm -amdgpu-kernarg-preload-count=16"
[ "$UNIT" = attention ] && FLAGS="$FLAGS -fno-honor-nans -fno-slp-vectorize"
hipcc $FLAGS $EXTRA -c $UNIT.hip -o /tmp/${UNIT}_${TAG}.o
OBJS=""
for o in gemm norm elementwise kdloss attention sampler prof comm api_ops model api_model; do
  if [ "$o" = "$UNIT" ]; then OBJS="$OBJS /tmp/${UNIT}_${TAG}.o"; else OBJS="$OBJS $o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpea_hip_${TAG}.so $OBJS -ldl
echo built ../libpea_hip_${TAG}.so

#!/bin/bash
# scripts/r04/build_variant.sh NAME FILE.hip "-DX -DY": frankensearch_amd/libfsgpu_variant_NAME.so = the default objects with FILE.hip
# recompiled under the extra definitions (same-box A/B runs copy a variant over libfsgpu.so; scripts/r03/ab_libs.sh)
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; defs=$3
python -m frankensearch_amd.build >/dev/null
obj=/tmp/fsgpu_variant_${name}_$(basename ${src%.*}).o
extra=""; [ "$src" = mfma_wide.hip ] && extra="-mllvm -pragma-unroll-threshold=200000"; [ "$src" = bert_kernels.hip -o "$src" = bert_query_kernels.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -x hip $extra $defs \
    -I include -c frankensearch_amd/csrc/$src -o $obj
objs=$(ls frankensearch_amd/_build/*.o | grep -v "/$(basename ${src%.*}).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o frankensearch_amd/libfsgpu_variant_${name}.so $objs $obj -ldl -pthread
echo frankensearch_amd/libfsgpu_variant_${name}.so

#!/bin/bash
# scripts/r05/build_variant_multi.sh NAME "-DX -DY" FILE1 FILE2 ...: frankensearch_amd/libfsgpu_variant_NAME.so = the default objects with the
# listed sources recompiled under the extra definitions (the three vector_index translation units share inline state: build them together).
set -e
cd "$(dirname "$0")/../.."
name=$1; defs=$2; shift 2
python -m frankensearch_amd.build >/dev/null
objs=$(ls frankensearch_amd/_build/*.o)
extra_objs=""
for src in "$@"; do
  obj=/tmp/fsgpu_variant_${name}_$(basename ${src%.*}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -x hip $defs \
      -I include -c frankensearch_amd/csrc/$src -o $obj
  objs=$(echo "$objs" | grep -v "/$(basename ${src%.*}).o")
  extra_objs="$extra_objs $obj"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o frankensearch_amd/libfsgpu_variant_${name}.so $objs $extra_objs -ldl -pthread
echo frankensearch_amd/libfsgpu_variant_${name}.so

#!/bin/bash
# build an A/B variant of the library: one translation unit recompiled with extra -D flags, linked with the objects of the regular build
#   scripts/build_variant.sh <name> <source.hip> [-DFOO=1 ...]   ->  ab_libs/libvidtok_amd_<name>.so
set -e
name=$1; src=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/ab_libs/obj
extra=""
case $src in conv_ws2.hip|tblock_ws128.hip) extra="-fno-slp-vectorize";; esac
obj=$R/ab_libs/obj/${name}_${src%.*}.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra "$@" -x hip -c $R/vidtok_amd/csrc/$src -o $obj 2>/dev/null
objs=""
for o in $R/vidtok_amd/build/*.o; do
  if [ "$(basename $o)" == "${src%.*}.o" ]; then objs="$objs $obj"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/ab_libs/libvidtok_amd_${name}.so
echo "built ab_libs/libvidtok_amd_${name}.so"

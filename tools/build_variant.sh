#!/bin/bash
# usage: tools/build_variant.sh <tag> <source.hip> [-DNAME=VALUE ...]
# Same-box A/B builds: compiles ONE source of densebox_amd/csrc with extra defines and links it with the other objects of the
# product build into densebox_amd/csrc/variants/libdensebox_hip_<tag>.so; run with DBX_LIB=<that path>.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; src=$2; shift; shift
C=$R/densebox_amd/csrc; mkdir -p $C/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable "$@" -c $C/$src -o $C/variants/${src%.hip}_$tag.o
objs=""
for f in $C/*.hip; do b=$(basename $f .hip); if [ "$b.hip" == "$src" ]; then objs="$objs $C/variants/${b}_$tag.o"; else objs="$objs $C/$b.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/libdensebox_hip_$tag.so $objs
echo $C/variants/libdensebox_hip_$tag.so

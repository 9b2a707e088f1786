#!/bin/bash
# A variant build of the library for same-box A/B runs (ACEZ_LIB): one translation unit recompiled with extra flags, the others taken from
# the in-tree build.   bash tools/lib_variant.sh <unit without .hip> <out.so> <extra hipcc flags...>
unit=$1; out=$2; shift 2
B=acezero_amd/build
extra=""
case $unit in ransac_api|cloud_api) extra="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra "$@" -c acezero_amd/csrc/$unit.hip -o /tmp/variant_$unit.o || exit 1
objs=""
for o in $B/*.o; do if [ $(basename $o) = $unit.o ]; then objs="$objs /tmp/variant_$unit.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs && echo built $out

#!/bin/bash
# A second build of libacez.so with extra compiler flags for head_api.hip, for A/B timing on one box through ACEZ_LIB:
#   tools/lib_variant.sh tools/libacez_x.so -DACEZ_SEQ_UNBOUNDED=1 ; ACEZ_LIB=tools/libacez_x.so python bench.py --headline-only
set -e
out=$1; shift
cd "$(dirname "$0")/.."
python -m acezero_amd.build >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c acezero_amd/csrc/head_api.hip -o /tmp/head_api_variant.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" /tmp/head_api_variant.o acezero_amd/build/acez_common.o acezero_amd/build/encoder_api.o acezero_amd/build/ransac_api.o acezero_amd/build/cloud_api.o
echo built "$out"

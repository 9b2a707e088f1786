#!/bin/bash
# Timing-only ablation builds of conv12p_kernel (CPU): tools/c12_variants.sh -> tools/libacez_c12_<n>.so for C12_ABL = n; run with
#   ACEZ_LIB=tools/libacez_c12_1.so rocprofv3 --kernel-trace --stats -- python tools/bench_encoder.py 64
set -e
cd "$(dirname "$0")/.."
python -m acezero_amd.build >/dev/null
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DC12_ABL=$n -c acezero_amd/csrc/encoder_api.hip -o /tmp/encoder_api_c12_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libacez_c12_$n.so /tmp/encoder_api_c12_$n.o acezero_amd/build/acez_common.o acezero_amd/build/head_api.o acezero_amd/build/ransac_api.o acezero_amd/build/cloud_api.o
  echo built tools/libacez_c12_$n.so
done

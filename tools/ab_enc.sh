#!/bin/bash
# Same-box A/B of the encoder: alternates tools/bench_encoder.py between the in-tree library and ACEZ_LIB builds.  bash tools/ab_enc.sh rounds lib1.so ...
N=$1; shift
for i in $(seq 1 $N); do
  for which in new "$@"; do
    if [ $which = new ]; then unset ACEZ_LIB; else export ACEZ_LIB=$which; fi
    for F in 64 128; do echo -n "$which  "; timeout 120 python tools/bench_encoder.py $F 2>/dev/null | tail -1; done
  done
done

#!/bin/bash
# A/B of an encoder environment switch on ONE box: bash tools/enc_ab.sh VAR A B [frames]   (alternating runs of tools/bench_encoder.py)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for v in $2 $3; do
    echo -n "$1=$v  "; env $1=$v python tools/bench_encoder.py ${4:-64} 2>&1 | tail -1
  done
done

#!/bin/bash
# Encoder ablations on one box (timing only): bash tools/enc_ablate.sh
export ACEZ_LIB=${ACEZ_LIB:-diag}   # the ablation switches exist in the diagnostics build only (python -m acezero_amd.build --diag)
cd $GRAFT_REPO_ROOT
for v in 0 2 4 6 0 2 4; do echo -n "ACEZ_CONV_DBG=$v  "; ACEZ_CONV_DBG=$v python tools/bench_encoder.py 64 2>&1 | tail -1; done

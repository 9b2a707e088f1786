#!/bin/bash
# Per-kernel durations of the encoder under ablation switches: bash tools/enc_kstats.sh "0 6 2 4"
export ACEZ_LIB=${ACEZ_LIB:-diag}   # the ablation switches exist in the diagnostics build only (python -m acezero_amd.build --diag)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/prof_enc
mkdir -p $OUT
for v in ${1:-0 6}; do
  ACEZ_CONV_DBG=$v rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/t$v -o trace -- timeout 200 python $R/tools/bench_encoder.py 64 > $OUT/t$v.log 2>&1
  echo "== ACEZ_CONV_DBG=$v"; tail -n 1 $OUT/t$v.log
  f=$(find $OUT/t$v -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "acez" in r["Name"]:
        print("  %-60s calls %4s avg %9.1f us total %8.2f ms" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
rm -rf $OUT

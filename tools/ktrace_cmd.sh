#!/bin/bash
# average duration per kernel name of any command under rocprofv3 --kernel-trace:  bash tools/ktrace_cmd.sh <tag> <command...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
tag=$1; shift
D=$R/gpurun_out/ktc_$tag; rm -rf $D; mkdir -p $D
( cd $R && rocprofv3 --output-format csv --kernel-trace --stats -d $D -o t -- "$@" > $D/log.txt 2>&1 )
f=$(find $D -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if "acez" in r["Name"]:
        print("%-90s calls %5s avg %9.1f us" % (re.sub(r"^void ", "", r["Name"]).split("(")[0][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY

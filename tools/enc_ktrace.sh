#!/bin/bash
# per-dispatch kernel durations of one encoder pass (tools/bench_encoder.py F), for the in-tree library and for ACEZ_LIB builds:
#   bash tools/enc_ktrace.sh F [lib.so ...]   -> average us per launch position of the pass, last 8 passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
F=$1; shift
for which in new "$@"; do
  if [ $which = new ]; then unset ACEZ_LIB; else export ACEZ_LIB=$R/$which; fi
  D=$R/gpurun_out/ktr_$(basename $which .so)
  rm -rf $D; mkdir -p $D
  rocprofv3 --output-format csv --kernel-trace -d $D -o t -- timeout 200 python $R/tools/bench_encoder.py $F > $D/log.txt 2>&1
  python - "$D" "$which" <<'PY'
import csv, glob, sys, re, collections
d, which = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
tr = [r for r in csv.DictReader(open(f)) if "acez" in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"^void acez::", "", r["Kernel_Name"]).split("(")[0][:48] for r in tr]
# pass length = distance between conv12p launches
idx = [i for i, n in enumerate(names) if n.startswith("conv12p")]
per = idx[1] - idx[0]
tr = tr[idx[-8]:idx[-8] + 8 * per] if len(idx) > 8 else tr[idx[0]:]
tot = 0
for k in range(per):
    ds = [int(tr[i]["End_Timestamp"]) - int(tr[i]["Start_Timestamp"]) for i in range(k, len(tr) - len(tr) % per, per)]
    us = sum(ds) / len(ds) / 1e3; tot += us
    print("%-22s %2d %-50s %8.1f us" % (which[-22:], k, names[idx[-8] + k] if len(idx) > 8 else names[idx[0] + k], us))
print("%-22s total %.1f us" % (which[-22:], tot))
PY
done

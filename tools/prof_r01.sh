#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun):  bash tools/prof_r01.sh [quick]
#   pass 1  rocprofv3 --kernel-trace --stats           -> per-kernel durations
#   pass 2-4 rocprofv3 --pmc ... (separate passes)     -> SQ counters, FETCH_SIZE, WRITE_SIZE
# Summaries land in gpurun_out/prof_keep/ (copied to profiles/ by hand and committed).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="timeout 300 python $R/bench.py --steps 100 --warmup 20 --buffer-patches 2000000 --reg-frames 1024 --e2e-frames 256 --session-frames 0 --no-cpu-baseline"
if [ "$1" = "quick" ]; then QUICK=1; fi
OUT=$R/gpurun_out/prof
KEEP=$R/gpurun_out/prof_keep
mkdir -p $OUT $KEEP
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
if [ -z "$QUICK" ]; then
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
fi
python - <<'PY'
import csv, glob, collections, json, os
root = os.environ.get("GRAFT_REPO_ROOT", ".")
out, keep = root + "/gpurun_out/prof", root + "/gpurun_out/prof_keep"
summary = {}
def agg(pattern):
    for f in glob.glob(out + "/" + pattern + "/**/*counter_collection.csv", recursive=True):
        d = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])
            d[k][0] += float(r["Counter_Value"]); d[k][1] += 1
        for k, v in sorted(d.items()):
            if "acez" in k[0] or "ransac" in k[0]:
                summary.setdefault(k[0], {})[k[1]] = {"mean_per_launch": v[0] / v[1], "launches": v[1]}
agg("pmc_sq"); agg("pmc_fetch"); agg("pmc_write")
json.dump(summary, open(keep + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
for name, c in summary.items():
    if "rowgemm80_kernel<true, false, false, 0>" in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:   # the plain forward instantiation
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
        # (MI355X_MICROARCH.md, HBM section) -> doubled. WRITE_SIZE is uncalibrated and taken as is.
        b = (2 * c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024
        json.dump({"kernel": name, "bytes_per_launch": b, "fetch_kib_raw": c["FETCH_SIZE"]["mean_per_launch"],
                   "write_kib_raw": c["WRITE_SIZE"]["mean_per_launch"], "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md"},
                  open(keep + "/r01_rowgemm_hbm_traffic.json", "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
PY
find $OUT/trace -name "*stats*.csv" -exec cp {} $KEEP/ \;
cp $OUT/*.log $KEEP/ 2>/dev/null
rm -rf $OUT
cut -c1-150 $KEEP/trace_kernel_stats.csv | head -12

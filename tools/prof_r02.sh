#!/bin/bash
# Round-2 profiling recipe (run on the GPU box through gpurun):  bash tools/prof_r02.sh
#   pass 1   rocprofv3 --kernel-trace --stats on the HEADLINE leg only (bench.py --headline-only: warm-up + timed training steps,
#            nothing else in the process) -> per-kernel durations of the training step
#   pass 2-4 rocprofv3 --pmc (separate passes) on the same command -> SQ counters, FETCH_SIZE, WRITE_SIZE
#   pass 5   the same --headline-only run with ACEZ_CHAIN=1 (the row-persistent chain kernel) -> its kernel durations
#   pass 6   the same with ACEZ_SEQ=0 (per-layer rowgemm80 launches instead of the one-launch chains of rowseq_kernel, which is
#            the default since the end of round 2) -> kernel durations, and FETCH_SIZE / WRITE_SIZE of one rowgemm80 layer launch
# Summaries land in gpurun_out/prof_keep/ (copied to profiles/ by hand and committed). The RANSAC kernel has its own recipe
# (tools/prof_ransac.sh).
export ACEZ_LIB=${ACEZ_LIB:-diag}   # the ablation switches exist in the diagnostics build only (python -m acezero_amd.build --diag)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
CMD="timeout 300 python $R/bench.py --headline-only --steps 300 --warmup 30 --buffer-patches 2000000"
OUT=$R/gpurun_out/prof
KEEP=$R/gpurun_out/prof_keep
mkdir -p $OUT $KEEP
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
ACEZ_CHAIN=1 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace_chain -o trace -- $CMD > $OUT/trace_chain.log 2>&1
ACEZ_SEQ=0 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace_layers -o trace -- $CMD > $OUT/trace_layers.log 2>&1
if [ "$1" != "quick" ]; then
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
ACEZ_SEQ=0 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_layers -o pmc -- $CMD > $OUT/pmc_fetch_layers.log 2>&1
ACEZ_SEQ=0 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_layers -o pmc -- $CMD > $OUT/pmc_write_layers.log 2>&1
fi
python - <<'PY'
import csv, glob, collections, json, os, re
root = os.environ.get("GRAFT_REPO_ROOT", ".")
out, keep = root + "/gpurun_out/prof", root + "/gpurun_out/prof_keep"
summary = {}
def short(n):
    return re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n)).split("(")[0][-70:]
for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_fetch_layers", "pmc_write_layers"):
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        d = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            d[k][0] += float(r["Counter_Value"]); d[k][1] += 1
        for k, v in sorted(d.items()):
            if "acez" in k[0]:
                summary.setdefault(k[0], {})[k[1]] = {"mean_per_launch": v[0] / v[1], "launches": v[1]}
json.dump(summary, open(keep + "/r02_pmc_summary.json", "w"), indent=1, sort_keys=True)
for name, c in summary.items():
    if "rowgemm80_kernel<true, false, false, 0>" in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:   # the plain forward instantiation
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
        # (MI355X_MICROARCH.md, HBM section) -> doubled. WRITE_SIZE is uncalibrated and taken as is.
        b = (2 * c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024
        json.dump({"kernel": name, "bytes_per_launch": b, "fetch_kib_raw": c["FETCH_SIZE"]["mean_per_launch"],
                   "write_kib_raw": c["WRITE_SIZE"]["mean_per_launch"], "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md"},
                  open(keep + "/r02_rowgemm_hbm_traffic.json", "w"), indent=1)
    if "rowseq_kernel<false>" in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:   # the forward chain: 8 layers per launch
        b = (2 * c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024
        json.dump({"kernel": name, "bytes_per_launch": b, "layers_per_launch": 8, "bytes_per_layer": b / 8,
                   "fetch_kib_raw": c["FETCH_SIZE"]["mean_per_launch"], "write_kib_raw": c["WRITE_SIZE"]["mean_per_launch"],
                   "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md; default head (8 wide layers)"},
                  open(keep + "/r02_rowseq_hbm_traffic.json", "w"), indent=1)
PY
for d in trace trace_chain trace_layers; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $KEEP/r02_${d}_kernel_stats.csv; done
grep -h "metric" $OUT/trace.log > $KEEP/r02_headline_under_rocprof.json
grep -h "metric" $OUT/trace_chain.log > $KEEP/r02_headline_chain_under_rocprof.json
rm -rf $OUT
cut -c1-150 $KEEP/r02_trace_kernel_stats.csv | head -14
cut -c1-150 $KEEP/r02_trace_chain_kernel_stats.csv | head -8
cat $KEEP/r02_headline_under_rocprof.json $KEEP/r02_headline_chain_under_rocprof.json

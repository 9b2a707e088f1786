#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun): kernel trace + three PMC passes of the same command.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 100 --warmup 20 --buffer-patches 2000000 --reg-frames 1024 --no-cpu-baseline"
if [ "$1" = "quick" ]; then QUICK=1; fi
OUT=$R/gpurun_out/prof
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
if [ -z "$QUICK" ]; then
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
fi
cd $OUT && find . -type f | head -40; du -sh .; tail -3 $OUT/trace.log
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof"
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("==", f); print(open(f).read()[:3000])
def agg(pattern, label):
    for f in glob.glob(out + "/" + pattern + "/**/*counter_collection.csv", recursive=True):
        d = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            d[k][0] += float(r["Counter_Value"]); d[k][1] += 1
        print("==", label, f)
        for k, v in sorted(d.items()):
            print("%-42s %-34s mean %.4g n %d" % (k[0], k[1], v[0] / v[1], v[1]))
agg("pmc_sq", "SQ"); agg("pmc_fetch", "FETCH"); agg("pmc_write", "WRITE")
PY

# keep only the summaries (gpurun_out is capped at 64 MiB)
mkdir -p $R/gpurun_out/prof_keep
find $OUT/trace -name "*stats*.csv" -exec cp {} $R/gpurun_out/prof_keep/ \;
cp $OUT/*.log $R/gpurun_out/prof_keep/ 2>/dev/null
rm -rf $OUT

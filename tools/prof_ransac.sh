#!/bin/bash
# RANSAC kernel counters (run on the GPU box through gpurun):  bash tools/prof_ransac.sh [tag]
#   pass 0  tools/ransac_phases.py                     -> where the time goes (refinement / tries / hypotheses switched down)
#   pass 1  rocprofv3 --kernel-trace --stats           -> launch durations of ransac_kernel (2048 frames, 32 hyps x 16 tries)
#   pass 2,3 rocprofv3 --pmc (separate passes, 8 SQ slots each) -> issue / wait / instruction-mix counters
# Summary: gpurun_out/prof_keep/<tag>_ransac_pmc.json (copied to profiles/ by hand and committed).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
TAG=${1:-r02}
OUT=$R/gpurun_out/prof_ransac
KEEP=$R/gpurun_out/prof_keep
mkdir -p $OUT $KEEP
CMD="timeout 200 python $R/tools/ransac_sweep.py 2048"
(cd $R && timeout 300 python tools/ransac_phases.py 2048) > $KEEP/${TAG}_ransac_phases.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $OUT/pmc_a -o pmc -- $CMD > $OUT/pmc_a.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT -d $OUT/pmc_b -o pmc -- $CMD > $OUT/pmc_b.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -d $OUT/pmc_c -o pmc -- $CMD > $OUT/pmc_c.log 2>&1
TAG=$TAG python - <<'PY'
import csv, glob, collections, json, os, re
root = os.environ.get("GRAFT_REPO_ROOT", ".")
tag = os.environ["TAG"]
out, keep = root + "/gpurun_out/prof_ransac", root + "/gpurun_out/prof_keep"
summary = {}
def short(n):
    n = re.sub(r"^void ", "", re.sub(r"<[^()]*>(?=\()", "", n))
    return re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][-60:]
for sub in ("pmc_a", "pmc_b", "pmc_c"):
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        d = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            d[k][0] += float(r["Counter_Value"]); d[k][1] += 1
        for k, v in sorted(d.items()):
            if "ransac" in k[0]:
                summary.setdefault(k[0], {})[k[1]] = {"mean_per_launch": v[0] / v[1], "launches": v[1]}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ransac" in r["Name"]:
            summary.setdefault(short(r["Name"]), {})["kernel_stats"] = {k: r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs")}
c = summary.get("ransac_kernel", {})
g = lambda n: c.get(n, {}).get("mean_per_launch")
if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
    wc = g("SQ_WAVE_CYCLES")
    summary["derived"] = {
        "note": "SQ_* cycle counters are quad-cycles summed over waves (MI355X_MICROARCH.md); fractions of wave-resident time",
        "valu_issue_frac_of_wave_cycles": g("SQ_ACTIVE_INST_VALU") / wc,
        "any_issue_frac_of_wave_cycles": (g("SQ_ACTIVE_INST_ANY") or 0) / wc,
        "wait_any_frac": (g("SQ_WAIT_ANY") or 0) / wc,
        "wait_inst_any_frac": (g("SQ_WAIT_INST_ANY") or 0) / wc,
        "valu_insts_per_wave": g("SQ_INSTS_VALU") / g("SQ_WAVES") if g("SQ_WAVES") else None,
    }
json.dump(summary, open(keep + "/%s_ransac_pmc.json" % tag, "w"), indent=1, sort_keys=True)
# the digest of the sources this pass was measured on: bench.py quotes the stored instruction count only while it matches the running build
import sys
sys.path.insert(0, root)
import bench
json.dump({"source_digest": bench.src_digest(bench.RANSAC_SOURCES), "sources": list(bench.RANSAC_SOURCES),
           "note": "digest of the ransac kernel sources %s_ransac_pmc.json was measured on" % tag}, open(keep + "/%s_ransac_pmc.digest.json" % tag, "w"), indent=1)
print(json.dumps(summary, indent=1)[:5000])
PY
cp $OUT/trace.log $KEEP/${TAG}_ransac_sweep.log 2>/dev/null
for f in a b c; do tail -n 3 $OUT/pmc_$f.log; done
rm -rf $OUT
cat $KEEP/${TAG}_ransac_phases.log


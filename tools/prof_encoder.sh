#!/bin/bash
# Encoder counters (gpurun): bash tools/prof_encoder.sh [tag]  -> gpurun_out/prof_keep/<tag>_encoder_pmc.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
TAG=${1:-r02}
OUT=$R/gpurun_out/prof_enc
KEEP=$R/gpurun_out/prof_keep
mkdir -p $OUT $KEEP
CMD="timeout 200 python $R/tools/bench_encoder.py 64"
rocprofv3 -L > gpurun_out/${TAG}_counters_available.txt 2>&1   # scratch, not a committed profile
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d $OUT/pmc_a -o pmc -- $CMD > $OUT/pmc_a.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $OUT/pmc_b -o pmc -- $CMD > $OUT/pmc_b.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc_c -o pmc -- $CMD > $OUT/pmc_c.log 2>&1
TAG=$TAG python - <<'PY'
import csv, glob, collections, json, os, re
root = os.environ.get("GRAFT_REPO_ROOT", ".")
tag = os.environ["TAG"]
out, keep = root + "/gpurun_out/prof_enc", root + "/gpurun_out/prof_keep"
summary = {}
def short(n):
    return re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n)).split("(")[0][-70:]
for sub in ("pmc_a", "pmc_b", "pmc_c"):
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        d = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            d[k][0] += float(r["Counter_Value"]); d[k][1] += 1
        for k, v in sorted(d.items()):
            if "acez" in k[0]:
                summary.setdefault(k[0], {})[k[1]] = {"mean_per_launch": v[0] / v[1], "launches": v[1]}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "acez" in r["Name"]:
            summary.setdefault(short(r["Name"]), {})["kernel_stats"] = {k: r[k] for k in ("Calls", "TotalDurationNs", "AverageNs")}
json.dump(summary, open(keep + "/%s_encoder_pmc.json" % tag, "w"), indent=1, sort_keys=True)
for k, c in summary.items():
    g = lambda n: c.get(n, {}).get("mean_per_launch")
    if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        print(k[:60], "mfma_busy/gui %.3f" % (g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 1024) if g("GRBM_GUI_ACTIVE") else -1),
              "lds_idx_active", g("SQ_LDS_IDX_ACTIVE"), "bank_conf", g("SQ_LDS_BANK_CONFLICT"), "data_fifo_full", g("SQ_LDS_DATA_FIFO_FULL"),
              "cmd_fifo_full", g("SQ_LDS_CMD_FIFO_FULL"), "wait_inst_lds", g("SQ_WAIT_INST_LDS"), "wave_cycles", g("SQ_WAVE_CYCLES"), "wait_any", g("SQ_WAIT_ANY"),
              "wait_inst_any", g("SQ_WAIT_INST_ANY"), "gui", g("GRBM_GUI_ACTIVE"), c.get("kernel_stats"))
PY
for f in a b c; do tail -n 2 $OUT/pmc_$f.log; done
rm -rf $OUT

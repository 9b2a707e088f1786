#!/bin/bash
# Round-6 profiling recipe (the round-4 recipe under the new tag; + an fp16 trace) (run on the GPU box through gpurun):  bash tools/prof_r06.sh [quick]
# New in round 4: pass 0 calibrates FETCH_SIZE / WRITE_SIZE on the step's own access patterns (tools/pmc_calib.hip: known
# byte counts, buffers 4x the Infinity Cache) and the factors are stored WITH the profile; every kernel of the step gets its duration,
# counter bytes and MFMA busy fraction in <tag>_step_hbm_traffic.json, so that bench.py's roofline block can be recomputed from profiles/ alone.
# Everything is measured on the DEFAULT path (one-launch GEMM chains, fused step), i.e. on the kernels bench.py times:
#   pass 1   rocprofv3 --kernel-trace --stats on the headline leg (bench.py --headline-only: warm-up + timed steps only)
#   pass 2   the same with --pose-refinement mlp (the step every non-seed ace_zero round runs)
#   pass 3   --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
#                  SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE         (8 SQ slots + GRBM)
#   pass 4/5 --pmc FETCH_SIZE / --pmc WRITE_SIZE (they do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots")
# Counter passes are their own runs with --kernel-trace only (no --stats, no other trace domain).
# Output (gpurun_out/prof_keep/, copied to profiles/ and committed): r06_kernel_stats_*.csv, r06_pmc_summary.json,
# r06_step_hbm_traffic.json (what bench.py's roofline.traffic / mfma_busy_frac quote: it names the kernels and carries the digest
# of the library sources it was measured on -- bench.py nulls the fields when the running build differs).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
CMD="timeout 300 python $R/bench.py --headline-only --steps 300 --warmup 30 --buffer-patches 2000000"
CMDP="timeout 300 python $R/bench.py --headline-only --steps 100 --warmup 30 --buffer-patches 2000000"
OUT=$R/gpurun_out/prof6
KEEP=$R/gpurun_out/prof_keep
mkdir -p $OUT $KEEP
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace_mlp -o trace -- $CMD --pose-refinement mlp > $OUT/trace_mlp.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace_fp16 -o trace -- $CMD --dtype fp16 > $OUT/trace_fp16.log 2>&1
if [ "$1" != "quick" ]; then
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o pmc -- $CMDP > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMDP > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMDP > $OUT/pmc_write.log 2>&1
fi
if [ "$1" != "quick" ]; then
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/cal_fetch -o pmc -- $R/tools/pmc_calib.bin > $OUT/cal_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/cal_write -o pmc -- $R/tools/pmc_calib.bin > $OUT/cal_write.log 2>&1
fi
python - <<'PY'
import csv, glob, collections, json, os, re
root = os.environ.get("GRAFT_REPO_ROOT", ".")
out, keep = root + "/gpurun_out/prof6", root + "/gpurun_out/prof_keep"
import sys
sys.path.insert(0, root)
import bench
digest = bench.src_digest(bench.STEP_SOURCES)
summary = {}
def short(n):
    return re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n)).split("(")[0][-70:]
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
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
# ---- calibration: counter value (KiB) / known bytes, per access pattern
calib = {}
try:
    known = json.loads([l for l in open(out + "/cal_fetch.log") if l.startswith("{")][-1])
    for sub, cname, key in (("cal_fetch", "FETCH_SIZE", "read_bytes"), ("cal_write", "WRITE_SIZE", "write_bytes")):
        for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if r["Counter_Name"] == cname and k in known and key in known[k]:
                    calib.setdefault(k, {})[cname + "_KiB"] = float(r["Counter_Value"])
                    calib[k][key] = known[k][key]
                    calib[k]["bytes_per_counted_byte"] = known[k][key] / (float(r["Counter_Value"]) * 1024)
except Exception as e:   # noqa: BLE001
    calib = {"error": repr(e)[:200]}
# the factors applied below: LDS-DMA K-stage rows (rowgemm80 / rowseq / wgrad operands) for the GEMM kernels, the lane-contiguous stream for the rest
def fac(name, default):
    return calib.get(name, {}).get("bytes_per_counted_byte", default) if isinstance(calib.get(name), dict) else default
F_ROWS, F_STREAM = fac("calib_ldsdma_rows", 1.0), fac("calib_stream16", 2.0)
W_ROWS, W_STREAM = fac("calib_write_rows", 1.0), fac("calib_write16", 1.0)
summary["_calibration"] = calib
summary["_source_digest"] = digest
json.dump(summary, open(keep + "/r06_pmc_summary.json", "w"), indent=1, sort_keys=True)
# the digest of what bench.py quotes. FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced
# read stream (MI355X_MICROARCH.md, HBM section) -> doubled. WRITE_SIZE is uncalibrated and taken as is.
step = {"source_digest": digest, "sources": list(bench.STEP_SOURCES), "note": "rocprofv3 --pmc, separate passes, bench.py --headline-only (default path); bytes = FETCH_SIZE x "
        "fetch_factor + WRITE_SIZE x write_factor with the factors measured by tools/pmc_calib.hip in the same session (`calibration`): the LDS-DMA K-stage pattern for "
        "the GEMM kernels (rowseq, wgrad), the lane-contiguous stream for the others; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)",
        "calibration": {"kernels": calib, "fetch_factor_gemm": F_ROWS, "fetch_factor_stream": F_STREAM, "write_factor_gemm": W_ROWS, "write_factor_stream": W_STREAM}, "kernels": {}}
for name, c in summary.items():
    if not isinstance(c, dict):
        continue
    g = lambda n: c.get(n, {}).get("mean_per_launch")
    e = {}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        gemm = ("rowseq_kernel" in name) or ("wgrad_kernel" in name) or ("wgrad_opt_kernel" in name) or ("rowgemm" in name)
        ff, wf = (F_ROWS, W_ROWS) if gemm else (F_STREAM, W_STREAM)
        e.update(bytes_per_launch=(ff * g("FETCH_SIZE") + wf * g("WRITE_SIZE")) * 1024, fetch_kib_raw=g("FETCH_SIZE"), write_kib_raw=g("WRITE_SIZE"),
                 fetch_factor=ff, write_factor=wf)
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("GRBM_GUI_ACTIVE"):
        e.update(mfma_busy_frac=g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 1024))
    if g("SQ_WAIT_ANY") is not None and g("SQ_WAVE_CYCLES"):
        e.update(wait_any_frac_of_wave_cycles=g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"))
    if g("SQ_BUSY_CU_CYCLES") is not None and g("GRBM_GUI_ACTIVE"):
        e.update(busy_cu_cycles=g("SQ_BUSY_CU_CYCLES"), gui_active=g("GRBM_GUI_ACTIVE"))
    if "kernel_stats" in c:
        e.update(avg_ns=float(c["kernel_stats"]["AverageNs"]), calls=int(c["kernel_stats"]["Calls"]))
    if e:
        step["kernels"][name] = e
for k in list(step["kernels"]):
    if "rowseq_kernel<false" in k and "F16" not in k and "bytes_per_launch" in step["kernels"][k]:
        step["kernels"][k].update(layers_per_launch=8, bytes_per_layer=step["kernels"][k]["bytes_per_launch"] / 8)
    if "rowseq_kernel<true" in k and "F16" not in k and "bytes_per_launch" in step["kernels"][k]:
        step["kernels"][k].update(layers_per_launch=7, bytes_per_layer=step["kernels"][k]["bytes_per_launch"] / 7)
json.dump(step, open(keep + "/r06_step_hbm_traffic.json", "w"), indent=1, sort_keys=True)
for k, e in sorted(step["kernels"].items()):
    print(k[:58].ljust(58), {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items() if a in ("avg_ns", "mfma_busy_frac", "wait_any_frac_of_wave_cycles", "bytes_per_launch")})
PY
for d in trace trace_mlp trace_fp16; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $KEEP/r06_kernel_stats_rocprofv3_headline_only_${d}.csv; done
grep -h "metric" $OUT/trace.log > $KEEP/r06_headline_under_rocprof.json
grep -h "metric" $OUT/trace_mlp.log > $KEEP/r06_headline_mlp_under_rocprof.json
grep -h "metric" $OUT/trace_fp16.log > $KEEP/r06_headline_fp16_under_rocprof.json
for f in pmc_sq pmc_fetch pmc_write; do [ -f $OUT/$f.log ] && tail -n 2 $OUT/$f.log; done
rm -rf $OUT
cut -c1-150 $KEEP/r06_kernel_stats_rocprofv3_headline_only_trace.csv | head -12
cut -c1-150 $KEEP/r06_kernel_stats_rocprofv3_headline_only_trace_mlp.csv | head -14
cat $KEEP/r06_headline_under_rocprof.json $KEEP/r06_headline_mlp_under_rocprof.json

#!/bin/bash
# Round-6 per-LAYER profile of the encoder and of the head's whole-frame inference (gpurun): bash tools/prof_r06_encoder.sh
#   pass 1  rocprofv3 --kernel-trace (per-dispatch start / end), pass 2 --pmc FETCH_SIZE, pass 3 --pmc WRITE_SIZE (separate runs),
# of (a) tools/bench_encoder.py 64 and (b) tools/head_maps_pass.py 64. Several layers share a kernel name, so the dispatches are told apart
# by their position in the pass (8 launches per encoder pass since round 6 -- res1_conv2 and res2_skip ride inside conv3x3r launches --, 1 per head pass: the wide layers, fc3 and the de-homogenisation are one launch since round 6) -> profiles/r06_encoder_layers.json: per layer us, HBM-side
# bytes (FETCH_SIZE x 2 x 1024 / 2... see the factors below: FETCH_SIZE is in KiB and counts half of the bytes on gfx950, WRITE_SIZE all of
# them: profiles/r04_step_hbm_traffic.json "calibration"), TFLOP/s and TB/s.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/prof6e
KEEP=$R/gpurun_out/prof_keep
mkdir -p $OUT $KEEP
for what in enc head; do
  if [ $what = enc ]; then CMD="timeout 200 python $R/tools/bench_encoder.py 64"; else CMD="timeout 200 python $R/tools/head_maps_pass.py 64 6"; fi
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${what}_trace -o t -- $CMD > $OUT/${what}_trace.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/${what}_fetch -o t -- $CMD > $OUT/${what}_fetch.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/${what}_write -o t -- $CMD > $OUT/${what}_write.log 2>&1
  tail -n 1 $OUT/${what}_trace.log
done
python - <<'PY'
import csv, glob, json, os, re, collections
root = os.environ.get("GRAFT_REPO_ROOT", ".")
out, keep = root + "/gpurun_out/prof6e", root + "/gpurun_out/prof_keep"
ENC = ["conv12 (conv1 1->32 + conv2 32->64 s2)", "conv3 64->128 s2", "conv4 128->256 s2", "res1_conv1 3x3 256->256 + res1_conv2 1x1 256->256 (back to back)",
       "res1_conv3 3x3 256->256 +res", "res2_conv1 3x3 256->512", "res2_conv2 1x1 512->512", "res2_conv3 3x3 512->512 + res2_skip 1x1 256->512 (extra K stages)"]
ENC_FLOP = [2 * 480 * 640 * 32 * 9 + 2 * 240 * 320 * 64 * 288, 2 * 120 * 160 * 128 * 576, 2 * 60 * 80 * 256 * 1152, 2 * 4800 * 256 * 2304 + 2 * 4800 * 256 * 256,
            2 * 4800 * 256 * 2304, 2 * 4800 * 512 * 2304, 2 * 4800 * 512 * 512, 2 * 4800 * 512 * 4608 + 2 * 4800 * 512 * 256]
HEAD = ["head: eight wide layers 512->512 + fc3 + de-homogenisation as one launch (head_maps_kernel)"]
HEAD_FLOP = [8 * 2 * 4800 * 512 * 512 + 2 * 4800 * 512 * 4]
res = {}
for what, names, flops in (("enc", ENC, ENC_FLOP), ("head", HEAD, HEAD_FLOP)):
    per = len(names)
    def rows(sub, pat):
        f = glob.glob(out + "/%s_%s/**/*%s.csv" % (what, sub, pat), recursive=True)
        return list(csv.DictReader(open(f[0]))) if f else []
    tr = [r for r in rows("trace", "kernel_trace") if "acez" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"] and "recast" not in r["Kernel_Name"]]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    npass = len(tr) // per
    tr = tr[len(tr) - npass * per:]
    dur = collections.defaultdict(list); kn = {}
    for i, r in enumerate(tr[2 * per:] if npass > 3 else tr):
        dur[i % per].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])); kn[i % per] = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][:60]
    cnt = {}
    for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        cr = [r for r in rows(sub, "counter_collection") if "acez" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"] and "recast" not in r["Kernel_Name"] and r["Counter_Name"] == cname]
        cr.sort(key=lambda r: int(r["Dispatch_Id"]))
        cr = cr[len(cr) - (len(cr) // per) * per:]
        d = collections.defaultdict(list)
        for i, r in enumerate(cr):
            d[i % per].append(float(r["Counter_Value"]))
        cnt[cname] = {k: sum(v) / len(v) for k, v in d.items()}
    lay = []
    for i in range(per):
        us = sum(dur[i]) / max(len(dur[i]), 1) / 1e3
        fb = cnt["FETCH_SIZE"].get(i, 0.0) * 1024 * 2.0      # KiB, x 2 (gfx950: the counter sees half of the bytes; calibrated in round 4)
        wb = cnt["WRITE_SIZE"].get(i, 0.0) * 1024
        lay.append({"layer": names[i], "kernel": kn.get(i), "us_per_64_frames": us, "fetch_MB": fb / 1e6, "write_MB": wb / 1e6,
                    "TBps": (fb + wb) / us / 1e6 if us else None, "TFLOPs": flops[i] * 64 / us / 1e6 if us else None,
                    "frac_of_mfma_peak": flops[i] * 64 / us / 1e6 / 2500.0 if us else None})
    tot = sum(x["us_per_64_frames"] for x in lay)
    res[what] = {"passes_averaged": len(dur[0]), "sum_us_per_64_frames": tot, "layers": lay}
    print(what, "sum %.1f us per 64 frames" % tot)
    for x in lay:
        print("  %-44s %-44s %8.1f us  fetch %7.1f MB  write %7.1f MB  %5.2f TB/s  %7.1f TFLOP/s" % (x["layer"], x["kernel"], x["us_per_64_frames"], x["fetch_MB"], x["write_MB"], x["TBps"] or 0, x["TFLOPs"] or 0))
json.dump(res, open(keep + "/r06_encoder_layers.json", "w"), indent=1)
PY
for what in enc head; do f=$(find $OUT/${what}_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $KEEP/r06_kernel_stats_rocprofv3_${what}.csv; done
rm -rf $OUT

"""Scale check of the in-process ACE0 loop: N synthetic frames, the reference's default iteration budgets (ace_zero.py:41-177),
mild augmentation (the stand-in encoder has no learned invariances).  python tools/reconstruct_synth.py [frames] [arc_deg]

Focal study: every round is printed with its focal NEXT TO the pose error after a similarity alignment of the
camera centres to the generator's poses (scale, median centre error relative to the trajectory extent, median rotation error), so
that a drifting focal can be read against what it does to the geometry. ACEZ_STUDY_FIXED_FOCAL=1 repeats the run with
refine_calibration off (focal pinned at the generator's) for the comparison."""
import json
import logging
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from acezero_amd import synth
from acezero_amd.session import ReconstructionSession, default_options

logging.basicConfig(level=logging.INFO)


from tools.pose_geometry import geometry


pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(pos[0]) if len(pos) > 0 else 1000
arc = float(pos[1]) if len(pos) > 1 else 0.36 * n
t0 = time.time()
seq = synth.render_room_sequence(seed=2089, n_frames=n, arc_deg=arc, device="cuda")
torch.cuda.synchronize()
t_render = time.time() - t0
esd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights_bandpass(seed=4099).items()}
import os
fixed = os.environ.get("ACEZ_STUDY_FIXED_FOCAL", "0") == "1"
opt = default_options(use_external_focal_length=seq["focal"], aug_rotation=2, aug_scale=1.06, aug_black_white=0.02, refine_calibration=not fixed)
t0 = time.time()
if os.environ.get("ACEZ_STUDY_LEVELS"):      # (study knobs: frames per encoder pass, augmentation scale levels)
    ReconstructionSession.AUG_SCALE_LEVELS = int(os.environ["ACEZ_STUDY_LEVELS"])
ses = ReconstructionSession(esd, seq["images"], opt=opt, depth=seq["depth"], chunk=int(os.environ.get("ACEZ_STUDY_CHUNK", "32")))
res = ses.reconstruct()
torch.cuda.synchronize()
dt = time.time() - t0
gt = seq["poses"].cpu().numpy().astype(np.float64)
ok = res["confidence"] > opt.registration_confidence
rounds = []
for h in res["history"]:
    r = {k: v for k, v in h.items() if k in ("id", "registration_rate", "focal", "mapped_images", "iterations", "map_seconds", "refit", "seed_rates")}
    okh = h["confidence"] > opt.registration_confidence
    a = r["aligned"] = geometry(h["poses"][okh], gt[okh]) if okh.sum() >= 3 else None
    rounds.append(r)
    if a:
        print(f"# {r['id']:>18}: focal {r['focal']:7.1f} ({r['focal'] / seq['focal']:.3f} x generator), registered {r['registration_rate'] * 100:5.1f}%, "
              f"alignment scale {a['scale']:.3f}, centre error {a['centre_rel_median'] * 100:.2f}% of the extent (p90 {a['centre_rel_p90'] * 100:.2f}%), "
              f"rotation {a['rot_abs_deg_median']:.2f} deg (rotation-only alignment), consecutive frames {a['rot_rel_deg_median']:.3f} deg")
out = {"frames": n, "arc_deg": arc, "render_s": t_render, "reconstruction_s": dt, "registered": float(ok.mean()), "focal": res["focal"],
       "refine_calibration": not fixed, "rounds": rounds, "timings": res.get("timings"), "compute_dtype": ses.dtype,
       "gpu_mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30}
print(json.dumps(out))
if "--store" in sys.argv:   # the stored figure bench.py quotes while the product sources are unchanged (bench.stored_session)
    import bench
    out["source_digest"] = bench.product_digest()
    out["command"] = "python tools/reconstruct_synth.py %d %g --store" % (n, arc)
    keep = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "prof_keep")
    os.makedirs(keep, exist_ok=True)
    tag = bench.PROFILE_TAG + "_session_1000_frames" + ("_" + ses.dtype if ses.dtype != "bf16" else "")
    json.dump(out, open(os.path.join(keep, tag + ".json"), "w"), indent=1)

"""Where does the final refit of the in-process ACE0 loop lose the geometry? (VERDICT r2 item 9 follow-up.)

tools/reconstruct_synth.py showed the loop's rounds 1-6 consistent with the generator's poses (centres within 1-3 % of the trajectory
extent after a similarity alignment) and the refit round far off (15 %), with or without focal refinement. This script runs the loop
up to the refit ONCE, then repeats the refit mapping (ace_zero_util.get_refit_mapping_cmd) in variants from the same state:
    python tools/refit_study.py [frames] [arc_deg]
Per variant: focal, registration rate, similarity-aligned centre error, rotation error after a rotation-only alignment, consecutive-
frame relative rotation error (gauge free), and how far the refiner moved the mapping poses away from the poses it was given."""
import json
import logging
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from acezero_amd import synth
from acezero_amd.session import ReconstructionSession, default_options

logging.basicConfig(level=logging.INFO)


from tools.pose_geometry import geometry


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
arc = float(sys.argv[2]) if len(sys.argv) > 2 else 144.0
seq = synth.render_room_sequence(seed=2089, n_frames=n, arc_deg=arc, device="cuda")
esd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights_bandpass(seed=4099).items()}
opt = default_options(use_external_focal_length=seq["focal"], aug_rotation=2, aug_scale=1.06, aug_black_white=0.02, final_refit=False)
ses = ReconstructionSession(esd, seq["images"], opt=opt, depth=seq["depth"])
res = ses.reconstruct()
gt = seq["poses"].cpu().numpy().astype(np.float64)
last = res["history"][-1]
poses, conf, focal = last["poses"], last["confidence"], last["focal"]
ok = conf > opt.registration_confidence
print("# state before the refit:", last["id"], "focal", round(focal, 1), json.dumps(geometry(poses[ok], gt[ok])))
sel = np.flatnonzero(conf >= opt.registration_confidence)
o = opt
base = dict(iterations=o.refit_iterations, loss_type="dyntanh", schedule="circle", lr_max=0.005, pose_wait=o.final_refit_posewait,
            refinement=o.refinement, refine_calibration=o.refine_calibration)
variants = [("refit as the reference runs it", {}),
            ("refit, poses fixed (refinement none)", dict(refinement="none")),
            ("refit, poses fixed, focal fixed", dict(refinement="none", refine_calibration=False)),
            ("refit, refinement from the first iteration", dict(pose_wait=0)),
            ("refit with the base round's loss and schedule", dict(loss_type=o.repro_loss_type, schedule=o.learning_rate_schedule, lr_max=o.learning_rate_max)),
            ("refit, warm start from the last map", dict(load_weights=last["head"])),
            ("refit, 5000 iterations", dict(iterations=5000, pose_wait=1000))]
only = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else range(len(variants))
for vi in only:
    name, kw = variants[vi]
    a = dict(base); a.update(kw)
    m = ses.map(sel, torch.from_numpy(poses[sel]), focal, tag=f"variant{vi}", **a)
    p2, c2 = ses.register(m["head"], m["focal"], tag=f"variant{vi}")
    ok2 = c2 > opt.registration_confidence
    refined_c2w = np.linalg.inv(np.concatenate([m["poses_w2c"].astype(np.float64), np.tile([[[0, 0, 0, 1.0]]], (len(sel), 1, 1))], 1))
    out = {"variant": name, "focal": round(m["focal"], 1), "batch_inliers": round(m["batch_inliers"], 3), "iterations": m["iterations"],
           "registered": round(float(ok2.mean()), 3), "registration_vs_generator": geometry(p2[ok2], gt[ok2]),
           "refined_mapping_poses_vs_given": geometry(refined_c2w, poses[sel]), "refined_mapping_poses_vs_generator": geometry(refined_c2w, gt[sel]),
           "registration_vs_refined_mapping_poses": geometry(p2[sel], refined_c2w)}
    print("# " + json.dumps(out))

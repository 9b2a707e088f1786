"""Exploration: mapping with known poses + relocalisation of held-out frames, then the full ACE0 loop, on the synthetic room."""
import logging
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from acezero_amd import synth
from acezero_amd.session import ReconstructionSession, default_options

logging.basicConfig(level=logging.INFO)


def pose_err(est, gt):
    dt = np.linalg.norm(est[:, :3, 3] - gt[:, :3, 3], axis=1)
    R = np.einsum("nij,nkj->nik", est[:, :3, :3], gt[:, :3, :3])
    ang = np.degrees(np.arccos(np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1, 1)))
    return dt, ang


n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
it = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
t0 = time.time()
seq = synth.render_room_sequence(seed=2089, n_frames=n, arc_deg=float(sys.argv[4]) if len(sys.argv) > 4 else 80.0, device="cuda")
torch.cuda.synchronize()
print("render", time.time() - t0)
esd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights_bandpass(seed=4099).items()}
opt = default_options(use_external_focal_length=seq["focal"], try_seeds=2, seed_iterations=it, iterations=it, refit_iterations=it,
                      iterations_max=8, final_refit_posewait=it // 5, learning_rate_warmup_iterations=min(1000, it // 5),
                      cooldown_iterations=it // 5, aug_rotation=float(sys.argv[5]) if len(sys.argv) > 5 else 15, aug_scale=float(sys.argv[6]) if len(sys.argv) > 6 else 1.5,
                      aug_black_white=0.02)
ses = ReconstructionSession(esd, seq["images"], opt=opt, depth=seq["depth"])
gt = seq["poses"].cpu().numpy()
even = list(range(0, n, 2))
m = ses.map(even, seq["poses"][even].cpu(), seq["focal"], iterations=it, loss_type="tanh", schedule="1cyclepoly", lr_max=0.003, tag="known-poses")
print({k: v for k, v in m.items() if k not in ("head", "poses_w2c")})
poses, inl = ses.register(m["head"], seq["focal"])
dt, ang = pose_err(poses, gt)
odd = np.arange(1, n, 2)
print("mapped frames: median", np.median(dt[even]), np.median(ang[even]), "inl", np.median(inl[even]))
print("held-out     : median", np.median(dt[odd]), np.median(ang[odd]), "max", dt[odd].max(), ang[odd].max(), "inl", np.median(inl[odd]), inl[odd].min())
if len(sys.argv) > 3 and sys.argv[3] == "full":
    res = ses.reconstruct()
    for h in res["history"]:
        print(h)
    # the estimated poses live in the seed camera's frame: compose with the seed's ground-truth pose
    np.random.seed(opt.random_seed)
    seeds = np.random.uniform(size=opt.try_seeds)
    best = int(res["history"][0]["id"].split("seed")[1])
    s = int(seeds[best] * n)
    est_world = np.einsum("ij,njk->nik", gt[s], res["poses"])
    dt, ang = pose_err(est_world, gt)
    ok = res["confidence"] > opt.registration_confidence
    print("ACE0: registered", ok.mean(), "median err", np.median(dt[ok]), np.median(ang[ok]), "max", dt[ok].max(), ang[ok].max(), "focal", res["focal"],
          "seconds", res["seconds"])

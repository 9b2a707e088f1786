"""Scale check of the in-process ACE0 loop: N synthetic frames, the reference's default iteration budgets (ace_zero.py:41-177),
mild augmentation (the stand-in encoder has no learned invariances).  python tools/reconstruct_synth.py [frames] [arc_deg]"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
arc = float(sys.argv[2]) if len(sys.argv) > 2 else 0.36 * n
t0 = time.time()
seq = synth.render_room_sequence(seed=2089, n_frames=n, arc_deg=arc, device="cuda")
torch.cuda.synchronize()
t_render = time.time() - t0
esd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights_bandpass(seed=4099).items()}
opt = default_options(use_external_focal_length=seq["focal"], aug_rotation=2, aug_scale=1.06, aug_black_white=0.02)
t0 = time.time()
ses = ReconstructionSession(esd, seq["images"], opt=opt, depth=seq["depth"])
res = ses.reconstruct()
torch.cuda.synchronize()
dt = time.time() - t0
gt = seq["poses"].cpu().numpy().astype(np.float64)
ok = res["confidence"] > opt.registration_confidence
out = {"frames": n, "arc_deg": arc, "render_s": t_render, "reconstruction_s": dt, "registered": float(ok.mean()), "focal": res["focal"],
       "rounds": [{k: v for k, v in h.items() if k in ("id", "registration_rate", "focal", "mapped_images", "iterations", "map_seconds", "refit", "seed_rates")}
                  for h in res["history"]],
       "gpu_mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30}
print(json.dumps(out))

"""Where the RANSAC kernel's time goes: the same 1024 frames with refinement / sampling tries / hypotheses switched down."""
import sys

import torch

sys.path.insert(0, ".")
from acezero_amd import _native as N
from acezero_amd import dsacstar, synth

fr = synth.make_registration_frames(seed=5, n_frames=64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = torch.from_numpy(fr["scene_coords"]).cuda().repeat(n // 64, 1, 1, 1).contiguous()
intr = [(fr["focal"], fr["ppx"], fr["ppy"])] * n
ids = list(range(n))
for name, hyps, tries, ref in (("full (32 hyps, 16 tries, refinement to convergence)", 32, 16, 100), ("one refinement round", 32, 16, 1),
                               ("two refinement rounds", 32, 16, 2), ("8 hypotheses, one round", 8, 16, 1), ("32 hypotheses, 1 try, one round", 32, 1, 1),
                               ("64 hypotheses, refinement", 64, 16, 100)):
    prm = N.RansacParams(hyps, tries, 10.0, 100.0, 100.0, 8, ref, 0)
    for _ in range(4):                                   # the first launches after a context (re)allocation are slow
        dsacstar.register_batch(sc, intr, prm, 1305, ids, want_masks=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        p, inl, _ = dsacstar.register_batch(sc, intr, prm, 1305, ids, want_masks=False)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / 3:8.3f} ms  mean inliers {float(inl.float().mean()):7.1f}")

"""RANSAC kernel time against the number of frames in a launch (one 256-thread workgroup per frame)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from acezero_amd import dsacstar, synth

fr = synth.make_registration_frames(seed=5, n_frames=64)
sc64 = torch.from_numpy(fr["scene_coords"]).cuda()
prm = dict(hyps=32, thr=10.0, alpha=100.0, max_reproj=100.0, sub=8, max_tries=16)
for n in [int(x) for x in sys.argv[1:]] or (32, 64, 128, 256, 384, 512, 768, 1024, 2048):
    sc = sc64.repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
    intr = [(fr["focal"], fr["ppx"], fr["ppy"])] * n
    dsacstar.register_batch(sc, intr, prm, 1305, list(range(n)), want_masks=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        dsacstar.register_batch(sc, intr, prm, 1305, list(range(n)), want_masks=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"frames {n:5d}  {ms:8.3f} ms  {n / ms:8.1f} frames/ms  per-frame-slot {ms / max(1, (n + 511) // 512):6.3f} ms")

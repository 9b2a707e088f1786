"""python tools/head_maps_pass.py [frames] [passes] -- acez_head_forward_maps on `frames` 60x80 frames of random features (the head's eight
512 -> 512 pointwise layers on whole frames + fc3): the launch sequence tools/prof_r05_encoder.sh profiles per layer."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from acezero_amd.head import HeadTrainer
from acezero_amd import synth, _native as N

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = F * 4800
tr = HeadTrainer(np.zeros(3, np.float32), max_batch=n, iterations=1, inference_only=True)
tr.load_flat(torch.from_numpy(synth.init_head_params(7)))
f = (torch.randn(n, 512, device="cuda") * 0.5).to(torch.bfloat16)
out = torch.empty((F, 3, 60, 80), dtype=torch.float32, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for p in range(P):
    if p == 2:
        e0.record()
    N.check(tr.lib.acez_head_forward_maps(tr._h, C.c_void_p(f.data_ptr()), F, 60, 80, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
e1.record()
torch.cuda.synchronize()
print("head on %d frames: %.3f ms per pass" % (F, e0.elapsed_time(e1) / (P - 2)))

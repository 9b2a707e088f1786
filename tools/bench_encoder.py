"""Encoder micro-benchmark: python tools/bench_encoder.py [frames] -- ms per frame and achieved TFLOP/s (58.4 GFLOP/frame at 480x640)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acezero_amd import synth
from acezero_amd.encoder import Encoder
from oracle import encoder_oracle

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sd = encoder_oracle.init_weights(seed=4099)
img = torch.from_numpy(synth.make_gray_images(seed=1, n=1, h=480, w=640)).cuda().repeat(F, 1, 1, 1).contiguous()
enc = Encoder(sd, max_frames=F, max_h=480, max_w=640)
out = torch.empty((F * 4800, 512), dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    enc.features_rows(img, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n):
    enc.features_rows(img, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
flop = 0
hw = [(480, 640, 1, 32, 9), (240, 320, 32, 64, 9), (120, 160, 64, 128, 9), (60, 80, 128, 256, 9), (60, 80, 256, 256, 9), (60, 80, 256, 256, 1),
      (60, 80, 256, 256, 9), (60, 80, 256, 512, 9), (60, 80, 512, 512, 1), (60, 80, 512, 512, 9), (60, 80, 256, 512, 1)]
for h, w, ci, co, k in hw:
    flop += 2 * h * w * ci * co * k
print("frames %d: %.3f ms/frame, %.1f frames/s, %.1f TFLOP/s (%.2f GFLOP/frame)" % (F, ms / F, F / ms * 1e3, flop * F / ms / 1e9, flop / 1e9))

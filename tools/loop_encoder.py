"""Runs the encoder in a loop for ~9 s (load generator for tools/clock_probe.sh)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acezero_amd import synth
from acezero_amd.encoder import Encoder
F = 64
sd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights(seed=4099).items()}
img = torch.from_numpy(synth.make_gray_images(seed=1, n=1, h=480, w=640)).cuda().repeat(F, 1, 1, 1).contiguous()
enc = Encoder(sd, max_frames=F, max_h=480, max_w=640)
out = torch.empty((F * 4800, 512), dtype=torch.bfloat16, device="cuda")
t0 = time.time()
n = 0
while time.time() - t0 < 9:
    for _ in range(20):
        enc.features_rows(img, out=out)
    torch.cuda.synchronize(); n += 20
print("frames/s", n * F / (time.time() - t0))

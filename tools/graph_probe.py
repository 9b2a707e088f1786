"""Does capturing the training step in a HIP graph shorten it?  One step captured with torch.cuda.CUDAGraph (the kernels are
launched through the C ABI on the capturing stream) and replayed, against the plain launch loop. Indices come from a fixed
staging buffer in both cases."""
import sys
import time

import torch

sys.path.insert(0, ".")
from acezero_amd import synth
from acezero_amd.head import HeadTrainer

B, steps = 5120, 600
prob = synth.make_training_problem(seed=3, n_images=60, views_per_image=10, patches_per_view=1024)
tr = HeadTrainer(prob["mean"], max_batch=B, loss_type="tanh", schedule="1cyclepoly", iterations=100000, lr_min=5e-4, lr_max=3e-3)
tr.load_flat(synth.init_head_params(1))
tr.set_buffer(prob["features"], prob["target_px"], prob["view_idx"], prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"], prob["view_image"],
              prob["image_pose_inv"])
n = tr.buffer_size
perm = torch.randperm(n, device=tr.device)
stage = perm[:B].clone()


def plain():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        stage.copy_(perm[(i % 100) * B:(i % 100 + 1) * B])
        tr.step(stage)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


for _ in range(20):
    tr.step(stage)
print("plain   us/step %.1f" % plain())
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        tr.step(stage)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    tr.step(stage)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    stage.copy_(perm[(i % 100) * B:(i % 100 + 1) * B])
    g.replay()
torch.cuda.synchronize()
print("graph   us/step %.1f" % ((time.perf_counter() - t0) / steps * 1e6))
print("plain   us/step %.1f" % plain())
st = tr.state()
print("iteration", st["iteration"], "loss", st["loss"])

"""Host-loop overhead of the training loop: us/step for different ways of feeding the per-epoch permutation."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from acezero_amd import synth
from acezero_amd.head import HeadTrainer, epoch_permutations

n, B, steps = 614400, 5120, 2400
prob = synth.make_training_problem(seed=3, n_images=60, views_per_image=10, patches_per_view=1024)
tr = HeadTrainer(prob["mean"], max_batch=B, loss_type="tanh", schedule="1cyclepoly", iterations=100000, lr_min=5e-4, lr_max=3e-3)
tr.load_flat(synth.init_head_params(1))
tr.set_buffer(prob["features"], prob["target_px"], prob["view_idx"], prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"], prob["view_image"],
              prob["image_pose_inv"])
n = tr.buffer_size
dev = tr.device


def run(kind, poll):
    gen = torch.Generator().manual_seed(1)
    perms = epoch_permutations(n, 1, dev) if kind == "device" else None
    fixed = torch.randperm(n).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    launched = 0
    while launched < steps:
        if kind == "naive":
            perm = torch.randperm(n, generator=gen).to(dev)
        elif kind == "device":
            perm = next(perms)
        else:
            perm = fixed
        for b0 in range(0, n - B + 1, B):
            tr.step(perm[b0:b0 + B])
            launched += 1
            if poll and launched % poll == 0:
                tr.state()
            if launched >= steps:
                break
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


for kind in ("fixed", "device", "fixed", "device", "naive"):
    for poll in (64,):
        print(kind, "poll", poll, "us/step %.1f" % run(kind, poll))

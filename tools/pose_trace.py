"""Timeline of the pose-refinement workgroups of the refinement step (diagnostics build, ACEZ_POSE_TRACE=1): s_memtime stamps of
thread 0 of every pose workgroup of the LAST step -- the forward (S3, inside step_begin_pose_kernel) and the reduce + backward chain
(S1, inside adamw_pose_kernel) -- relative to the workgroup's own entry stamp, in microseconds (TICK_US overrides the tick the tool infers).
  python tools/pose_trace.py          (on the GPU box)"""
import ctypes as C
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, ".")
os.environ["ACEZ_POSE_TRACE"] = "1"
from acezero_amd import _native as N

with N.diag_library():
    import bench
    args = types.SimpleNamespace(pose_refinement="mlp", steps=20, buffer_patches=2_000_000, warmup=10)
    from acezero_amd import synth
    from acezero_amd.head import HeadTrainer
    device = torch.device("cuda:0")
    prob, feats, target_px, view_idx = bench.make_buffer(2_000_000, device, 2089, n_images=int(os.environ.get("N_IMAGES", "1000")), grid=(80, 60))
    tr = HeadTrainer(prob["mean"], max_batch=5120, global_batch=5120, loss_type="tanh", schedule="1cyclepoly", iterations=25000, lr_min=0.0005,
                     lr_max=0.003, warmup_iterations=1000, warmup_lr=0.0005, cooldown_iterations=5000, pose_refinement="mlp", dtype="bf16",
                     refine_calibration=True, focal_init=float(prob["focal"]))
    tr.load_flat(torch.from_numpy(synth.init_head_params(1)))
    tr.set_buffer(feats, target_px, view_idx, prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"], prob["view_image"], prob["image_pose_inv"])
    perm = torch.randperm(2_000_000, device=device)
    batches = [perm[i * 5120:(i + 1) * 5120].contiguous() for i in range(41)]
    for i in range(40):
        tr.step(batches[i], batches[i + 1])
    torch.cuda.synchronize()
    raw = np.zeros(3 * 1024 * 16, np.uint64)
    N.check(tr.lib.acez_trainer_debug_read(tr._h, 8, 0, raw.ctypes.data_as(C.c_void_p), raw.nbytes, None))
t = raw.reshape(3, 1024, 16).astype(np.int64)
# s_memtime: the 100 MHz constant clock or the shader clock (~2.4 GHz), depending on the part's firmware: decided from the size of an
# S1 workgroup's entry -> exit difference (the launch lasts ~20 us)
_d = np.median((t[1][:, 8] - t[1][:, 0])[t[1][:, 0] > 0])
tick_us = float(os.environ.get("TICK_US", "0.01" if _d < 20000 else str(1 / 2400.0)))
print("ticks of an S1 workgroup:", _d, "-> tick =", tick_us, "us")
names = [["entry", "conv1", "conv2", "conv3", "skip + res", "fc1", "fc2", "fc3", "poses stored"],
         ["entry", "reduce done", "compose backward", "through fc3", "through fc2", "through fc1", "a1 mask requested", "through conv3", "through conv2",
          "(reduce) table scan", "(reduce) hit rows in LDS", "(compose) raw pose ready", "(compose) arithmetic done"],
         ["entry", "optimiser operands requested", "products done", "partial sums written", "barrier passed", "stored"]]
order = [[0, 1, 2, 3, 4, 5, 6, 7, 8], [0, 9, 10, 1, 11, 12, 2, 3, 4, 5, 6, 7, 8], [0, 1, 2, 3, 4, 5]]
for slot, title in enumerate(["S3: pose forward workgroups (step_begin_pose_kernel)", "S1: reduce + backward chain workgroups (adamw_pose_kernel)",
                              "S2: weight-gradient tiles + AdamW (pose_mlp_wgrad_kernel)"]):
    x = t[slot]
    ok = x[:, 0] > 0
    rel = (x[ok] - x[ok][:, :1]) * tick_us
    print(title, "--", int(ok.sum()), "workgroups")
    prev = np.zeros(int(ok.sum()))
    for i in order[slot]:
        v = rel[:, i]
        print(f"  {names[slot][i]:26s} median {np.median(v):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}   (+{np.median(v - prev):5.2f})")
        prev = v
# S2 by job class: layer (job_start of launch_pose_wgrad: 8, 64, 64, 8, 64, 64, 8 tiles) and whether the tile also produces the bias gradient
starts = [0, 8, 72, 136, 144, 208, 272, 280]
lname = ["fc3", "fc2", "fc1", "skip", "conv3", "conv2", "conv1"]
kb = [8, 8, 8, 1, 8, 8, 1]
x = t[2][:280]
tot = (x[:, 5] - x[:, 0]) * tick_us
for l in range(7):
    j = np.arange(starts[l], starts[l + 1])
    bias = ((j - starts[l]) % kb[l]) == 0
    for nm, sel in (("bias tiles", bias), ("other tiles", ~bias)):
        if sel.any():
            v = tot[j[sel]]
            print(f"  S2 {lname[l]:6s} {nm:11s} n {sel.sum():3d}  median {np.median(v):6.2f}  max {v.max():6.2f}")

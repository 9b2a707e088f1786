"""Stand-alone durations of the pose-refinement kernels (run under rocprofv3 --kernel-trace --stats on the GPU box):
    ACEZ_POSE_TILE=8 ACEZ_POSE_WB=32 python tools/pose_kernels_timing.py [n_images] [rows]
Drives the split flow (acez_train_backward / acez_train_update), in which the reduction + backward chain (pose_s1t_kernel) and the
weight gradients (pose_mlp_wgrad_kernel) are their own launches, and acez_trainer_get_poses (pose_fwd_t_kernel)."""
import os as _os
_os.environ.setdefault("ACEZ_LIB", "diag")   # the ACEZ_* ablation switches exist in the diagnostics build only (acezero_amd/build.py --diag)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 5120
    from acezero_amd import synth
    from acezero_amd.head import HeadTrainer
    dev = torch.device("cuda", 0)
    prob, feats, target_px, view_idx = bench.make_buffer(400_000, dev, 2089, n_images=n_images)
    tr = HeadTrainer(prob["mean"], max_batch=rows, loss_type="tanh", schedule="constant", iterations=10000, lr_min=0.0005, pose_refinement="mlp",
                     refine_calibration=True, focal_init=float(prob["focal"]))
    tr.load_flat(torch.from_numpy(synth.init_head_params(1)))
    tr.set_buffer(feats, target_px, view_idx, prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"], prob["view_image"], prob["image_pose_inv"])
    perm = torch.randperm(400_000, device=dev)
    for i in range(60):
        tr.backward(perm[i * rows:(i + 1) * rows].contiguous())
        tr.update()
        if i % 4 == 0:
            tr.current_poses()
    torch.cuda.synchronize()
    print("done", tr.state()["iteration"])


if __name__ == "__main__":
    main()

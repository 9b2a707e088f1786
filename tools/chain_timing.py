"""Step time of the training step at BASELINE's batch (5120 rows, 8 wide layers) with the one-launch chain kernel and with the
per-layer launches (ACEZ_CHAIN=0), same process, same buffer.  python tools/chain_timing.py [buffer_patches] [steps]"""
import os as _os
_os.environ.setdefault("ACEZ_LIB", "diag")   # the ACEZ_* ablation switches exist in the diagnostics build only (acezero_amd/build.py --diag)
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def run(chain, patches, steps, pose="none"):
    os.environ["ACEZ_CHAIN"] = chain
    from acezero_amd import synth
    from acezero_amd.head import HeadTrainer
    dev = torch.device("cuda", 0)
    prob, feats, target_px, view_idx = bench.make_buffer(patches, dev, 2089)
    tr = HeadTrainer(prob["mean"], max_batch=5120, loss_type="tanh", schedule="1cyclepoly", iterations=25000, lr_min=0.0005, lr_max=0.003,
                     warmup_iterations=1000, warmup_lr=0.0005, cooldown_iterations=5000, pose_refinement=pose,
                     refine_calibration=pose != "none", focal_init=float(prob["focal"]))
    tr.load_flat(torch.from_numpy(synth.init_head_params(1)))
    tr.set_buffer(feats, target_px, view_idx, prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"], prob["view_image"], prob["image_pose_inv"])
    g = torch.Generator(device=dev).manual_seed(8191)
    perm = torch.randperm(patches, generator=g, device=dev)
    batches = [perm[i * 5120:(i + 1) * 5120].contiguous() for i in range(patches // 5120)]
    for i in range(30):
        tr.step(batches[i % len(batches)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(batches[(30 + i) % len(batches)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tr.set_profiling(True)
    for i in range(20):
        tr.step(batches[i % len(batches)])
    torch.cuda.synchronize()
    prof = {k: round(1e3 * v[0] / 20, 1) for k, v in tr.get_profile().items() if v[1]}
    st = tr.state()
    print("chain=%s pose=%s: %.1f us/step  loss %.4f  classes(us/step) %s" % (chain, pose, dt * 1e6, st["loss"], prof), flush=True)
    tr.close()


if __name__ == "__main__":
    patches = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    if len(sys.argv) > 3:   # ablation sweep of the chain kernel (timing only: results are wrong with any switch set)
        for dbg in sys.argv[3].split(","):
            os.environ["ACEZ_CHAIN_DBG"] = dbg
            print("ACEZ_CHAIN_DBG=" + dbg, end="  ")
            run("1", patches, steps)
        sys.exit(0)
    for chain in ("1", "0"):
        run(chain, patches, steps)
    for chain in ("1", "0"):
        run(chain, patches, steps, pose="mlp")

"""Why does the forward chain take 43 us in one process and 46.5 us in the next (same binary, same box)? Re-creates the trainer several
times inside ONE process, with a differently sized dummy allocation in between, and prints the per-class times of each incarnation
next to the addresses of its weight / activation buffers' 2 MiB offsets.  python tools/bimodal_probe.py [rounds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def one(tag, patches=1_000_000, steps=300):
    from acezero_amd import synth
    from acezero_amd.head import HeadTrainer
    dev = torch.device("cuda", 0)
    prob, feats, target_px, view_idx = bench.make_buffer(patches, dev, 2089)
    tr = HeadTrainer(prob["mean"], max_batch=5120, loss_type="tanh", schedule="1cyclepoly", iterations=25000, lr_min=0.0005, lr_max=0.003,
                     warmup_iterations=1000, warmup_lr=0.0005, cooldown_iterations=5000)
    tr.load_flat(torch.from_numpy(synth.init_head_params(1)))
    tr.set_buffer(feats, target_px, view_idx, prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"], prob["view_image"], prob["image_pose_inv"])
    g = torch.Generator(device=dev).manual_seed(8191)
    perm = torch.randperm(patches, generator=g, device=dev)
    batches = [perm[i * 5120:(i + 1) * 5120].contiguous() for i in range(patches // 5120)]
    for i in range(30):
        tr.step(batches[i % len(batches)], batches[(i + 1) % len(batches)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(batches[(30 + i) % len(batches)], batches[(31 + i) % len(batches)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tr.set_profiling(True)
    for i in range(20):
        tr.step(batches[i % len(batches)], batches[(i + 1) % len(batches)])
    torch.cuda.synchronize()
    prof = {k: round(1e3 * v[0] / 20, 1) for k, v in tr.get_profile().items() if v[1]}
    print(f"{tag}: {dt * 1e6:.1f} us/step  {prof}  params@{tr.params.data_ptr() % (1 << 21):#x} feats@{feats.data_ptr() % (1 << 21):#x}", flush=True)
    tr.close()


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    keep = []
    for r in range(rounds):
        one(f"incarnation {r}")
        keep.append(torch.empty((r + 1) * 3_000_001, device="cuda"))   # shifts what the next incarnation's hipMalloc calls return

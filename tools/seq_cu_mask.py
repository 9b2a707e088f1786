"""rowseq_kernel under a restricted CU set (run on the GPU box; tests/test_seq_gpu.py runs it in a subprocess):

    HSA_CU_MASK=0:0-223 python tools/seq_cu_mask.py [steps] [rows]

With some CUs masked off the workgroup -> XCD mapping of a launch and the number of resident workgroups are no longer what the
one-launch GEMM chains assume. The trainer must then do one of three safe things, and this script checks that it did:
  * the placement probe of acez_trainer_create fails (or too few CUs are visible) -> per-layer launches from the start;
  * the chains run and every step is bit-identical to per-layer launches;
  * a hand-off poll expires -> the step is abandoned, the trainer falls back, and the surviving steps are bit-identical.
Prints one JSON line: {"probe", "enabled_at_start", "enabled_at_end", "faults", "effective_steps", "bit_identical", "cus"}."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make(prob, seq, iters, rows):
    from acezero_amd import synth
    from acezero_amd.head import HeadTrainer
    os.environ["ACEZ_SEQ"] = seq
    try:
        tr = HeadTrainer(prob["mean"], num_head_blocks=1, use_homogeneous=True, max_batch=rows, loss_type="tanh", schedule="constant",
                         iterations=iters, lr_min=3e-4)
    finally:
        os.environ.pop("ACEZ_SEQ", None)
    tr.load_flat(torch.from_numpy(synth.init_head_params(11, num_head_blocks=1, use_homogeneous=True)))
    tr.set_buffer(prob["features"], prob["target_px"], prob["view_idx"], prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"],
                  prob["view_image"], prob["image_pose_inv"])
    return tr


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 5120
    from acezero_amd import synth
    os.environ.setdefault("ACEZ_SEQ_SPIN_US", "5000")
    prob = synth.make_training_problem(seed=7, n_images=16, views_per_image=2, patches_per_view=256)
    new = make(prob, "1", steps + 4, rows)
    ref = make(prob, "0", steps + 4, rows)
    st0 = new.seq_status()
    rng = np.random.default_rng(3)
    n = prob["features"].shape[0]
    effective, it = 0, 0
    for s in range(steps):
        nrow = rows if s % 3 else max(80, rows // 3 + 7)     # full and ragged batches
        idx = torch.from_numpy(rng.permutation(n)[:nrow].astype(np.int64)).cuda()
        new.step(idx)
        now = new.state()["iteration"]                        # (a state read also performs a pending fall-back)
        if now == it + 1:
            ref.step(idx)
            effective += 1
        it = now
    torch.cuda.synchronize()
    same = bool(torch.equal(new.params, ref.params) and torch.equal(new.adam_m, ref.adam_m) and torch.equal(new.adam_v, ref.adam_v))
    st1 = new.seq_status()
    print(json.dumps({"probe": st0["probe"], "enabled_at_start": st0["enabled"], "enabled_at_end": st1["enabled"], "faults": st1["faults"],
                      "effective_steps": effective, "steps": steps, "bit_identical": same,
                      "cus": torch.cuda.get_device_properties(0).multi_processor_count, "mask": os.environ.get("HSA_CU_MASK", "")}))
    return 0 if same and effective >= steps - 2 * max(1, st1["faults"]) else 1


if __name__ == "__main__":
    sys.exit(main())

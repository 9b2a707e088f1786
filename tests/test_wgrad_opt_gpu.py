"""GPU: wgrad_opt_kernel (head_kernels.hip) -- the optimiser step of the wide layers, the small parameters and the schedule wave inside
the weight-gradient launch, the next batch gathered beside the loss kernel -- against the flow it replaces (ACEZ_WGRAD_OPT=0 in the
diagnostics build: wgrad_kernel + the optimiser launch). The two slabs of a tile are added in the same order, the small parameters are
reduced in tail_output's order and every element goes through the same adamw_one: EVERYTHING must agree bit for bit, step after step
(a hand-off read too early would show up as a diverging trajectory). Plus the fault path of its hand-off (ACEZ_WGO_FAULT_AT)."""
import os

import numpy as np
import pytest
import torch

from tests import helpers
from tests.helpers import big_problem as _big_problem
from tests.test_head_gpu import _trainer
from tests.test_seq_gpu import _small_trainer

pytestmark = pytest.mark.gpu


def _pair(make):
    out = []
    for opt in ("0", "1"):
        os.environ["ACEZ_WGRAD_OPT"] = opt
        try:
            out.append(make())
        finally:
            os.environ.pop("ACEZ_WGRAD_OPT", None)
    return out


@pytest.mark.parametrize("name,n,dtype", [("head_tanh_1cyclepoly", 5120, "bf16"), ("head_dyntanh_circle", 637, "bf16"),
                                          ("head_tanh_calib", 80, "bf16"), ("head_tanh_1cyclepoly", 4097, "fp16")])
def test_fused_launches_equal_the_two_launch_flow(name, n, dtype, diag_lib):
    prob = _big_problem()
    if dtype == "fp16":
        prob = dict(prob)
        prob["features"] = prob["features"].astype(np.float16).astype(np.float32)
    from oracle import head_oracle
    flat0 = head_oracle.init_params(helpers.SEED + 1)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS[name], prob)
    cfg["global_batch"] = n
    ref, new = _pair(lambda: _trainer(prob, flat0, cfg, max_batch=5120, dtype=dtype))
    rng = np.random.default_rng(9)
    batches = [torch.from_numpy(rng.permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda() for _ in range(12)]
    for tr in (ref, new):
        for i, idx in enumerate(batches):
            # a mix of announced and unannounced next batches, and one announcement that is not honoured (the trainer must gather again)
            if i % 4 == 3:
                tr.step(idx)
            elif i == 5:
                tr.step(idx, batches[0])
            else:
                tr.step(idx, batches[i + 1] if i + 1 < len(batches) else None)
        torch.cuda.synchronize()
    assert torch.equal(ref.params, new.params) and torch.equal(ref.adam_m, new.adam_m) and torch.equal(ref.adam_v, new.adam_v)
    assert ref.state() == new.state()
    f = torch.from_numpy(prob["features"][:777]).cuda()
    assert torch.equal(ref.get_scene_coordinates(f), new.get_scene_coordinates(f))
    lr, ln = ref.log(0, 12), new.log(0, 12)
    assert np.array_equal(lr[0], ln[0]) and np.array_equal(lr[1], ln[1])


@pytest.mark.parametrize("name,n,dtype", [("head_tanh_1cyclepoly", 5120, "bf16"), ("head_dyntanh_circle", 637, "bf16"),
                                          ("head_tanh_calib", 4097, "bf16"), ("head_tanh_1cyclepoly", 5120, "fp16"),
                                          ("head_dyntanh_circle", 4097, "fp16")])
def test_product_library_fused_step_equals_backward_plus_update(name, n, dtype):
    """The same comparison on libacez.so itself (no diagnostics build, no environment switch): the product can run the two-launch flow
    through backward() + update() -- wgrad_kernel, grad_reduce_kernel, adamw_kernel -- and step(idx, next) is the four-launch fused flow
    (wgrad_opt_kernel with the optimiser, the small parameters and the schedule wave inside, the next batch gathered beside the loss).
    Parameters, both moments, the schedule state and the logged losses must agree bit for bit at BASELINE's batch and at ragged ones."""
    prob = _big_problem()
    if dtype == "fp16":
        prob = dict(prob)
        prob["features"] = prob["features"].astype(np.float16).astype(np.float32)
    from oracle import head_oracle
    flat0 = head_oracle.init_params(helpers.SEED + 1)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS[name], prob)
    cfg["global_batch"] = n
    fused = _trainer(prob, flat0, cfg, max_batch=5120, dtype=dtype)
    split = _trainer(prob, flat0, cfg, max_batch=5120, dtype=dtype)
    assert "diag" not in os.path.basename(fused.lib._name) and fused.seq_status()["enabled"]
    rng = np.random.default_rng(13)
    batches = [torch.from_numpy(rng.permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda() for _ in range(10)]
    for i, idx in enumerate(batches):
        if i % 4 == 3:
            fused.step(idx)                       # an unannounced batch: gathered by the step itself
        else:
            fused.step(idx, batches[i + 1] if i + 1 < len(batches) else None)
        split.backward(idx)
        split.update()
    torch.cuda.synchronize()
    assert torch.equal(fused.params, split.params) and torch.equal(fused.adam_m, split.adam_m) and torch.equal(fused.adam_v, split.adam_v)
    a, b = fused.state(), split.state()
    assert a == b and a["iteration"] >= 9    # (one of the golden configurations ends its schedule inside the run: the last step is a no-op in both)
    lf, ls = fused.log(0, 10), split.log(0, 10)
    assert np.array_equal(lf[0], ls[0]) and np.array_equal(lf[1], ls[1])
    f = torch.from_numpy(prob["features"][:777]).cuda()
    assert torch.equal(fused.get_scene_coordinates(f), split.get_scene_coordinates(f))


@pytest.mark.parametrize("fault_mod,dtype,announce", [(0, "bf16", True), (3, "bf16", True), (3, "fp16", True), (3, "bf16", False), (0, "fp16", False)])
def test_expired_exchange_poll_finishes_the_step_in_the_fall_back(fault_mod, dtype, announce, diag_lib):
    """Fault injection (ACEZ_WGO_FAULT_AT, ACEZ_WGO_FAULT_MOD): in one launch every workgroup -- or every third -- waits for a partner
    count that never comes: what a tile whose two slabs sit on different XCDs looks like. No hang; no weight tile is stored from an
    incomplete sum. The small parameters and the schedule wave of that launch read the fault word long before it is raised, so the step
    IS applied in part: the fall-back therefore FINISHES it -- every launch that writes a step's buffers
    is a no-op while the fault word is up, wgrad_kernel recomputes the slabs from the untouched operands and wgo_recover_kernel applies
    the step's own AdamW scalars to exactly the rows whose wave gave up. After the state read the trainer is, bit for bit, the two-launch
    flow's trainer after the same steps: a step is atomic again (ace_trainer.py:620-640).
    announce = False: the faulting step and the steps queued behind it do NOT announce their successors, so their gathers are
    the plain gather launches into R[0] -- the faulted step's layer-0 operand -- which must hold while the fault word is up as well."""
    prob = _big_problem(n_images=8, patches_per_view=256)
    if dtype == "fp16":
        prob = dict(prob)
        prob["features"] = prob["features"].astype(np.float16).astype(np.float32)

    def make():
        from acezero_amd.head import HeadTrainer
        from acezero_amd import synth
        tr = HeadTrainer(prob["mean"], num_head_blocks=1, use_homogeneous=True, max_batch=2048, loss_type="tanh", schedule="1cyclepoly",
                         iterations=50, lr_min=1e-4, lr_max=6e-4, warmup_iterations=10, cooldown_iterations=10, dtype=dtype)
        tr.load_flat(torch.from_numpy(synth.init_head_params(11, num_head_blocks=1, use_homogeneous=True)))
        tr.set_buffer(prob["features"], prob["target_px"], prob["view_idx"], prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"],
                      prob["view_image"], prob["image_pose_inv"])
        return tr

    os.environ.update(ACEZ_WGO_FAULT_AT="1", ACEZ_SEQ_SPIN_US="3000", ACEZ_WGO_FAULT_MOD=str(fault_mod))
    try:
        new = make()
    finally:
        for k in ("ACEZ_WGO_FAULT_AT", "ACEZ_SEQ_SPIN_US", "ACEZ_WGO_FAULT_MOD"):
            os.environ.pop(k)
    os.environ["ACEZ_WGRAD_OPT"] = "0"
    try:
        ref = make()              # the two-launch flow (wgrad_kernel + the optimiser launch) throughout
    finally:
        os.environ.pop("ACEZ_WGRAD_OPT")
    rng = np.random.default_rng(4)
    batches = [torch.from_numpy(rng.permutation(prob["features"].shape[0])[:2048].astype(np.int64)).cuda() for _ in range(7)]
    new.step(batches[0], batches[1]); ref.step(batches[0], batches[1])
    torch.cuda.synchronize()
    assert torch.equal(new.params, ref.params)
    L = new.L
    wide = lambda t: torch.cat([t[l * 262656:l * 262656 + 262144] for l in range(L)])
    w0, m0 = wide(new.params).clone(), wide(new.adam_m).clone()
    if announce:
        new.step(batches[1], batches[2])      # the faulting step (its successor's gather rides beside its loss kernel, before the fault)
    else:
        new.step(batches[1])                  # the faulting step, no successor announced
    torch.cuda.synchronize()
    changed = int((wide(new.params) != w0).sum())
    if fault_mod == 0:
        assert changed == 0 and torch.equal(wide(new.adam_m), m0)     # every exchange timed out: no weight tile stored
    else:
        assert 0 < changed < w0.numel()                                # the tiles whose exchange completed were stored, the others not
    p1, m1, v1 = new.params.clone(), new.adam_m.clone(), new.adam_v.clone()
    if announce:
        new.step(batches[2], batches[3])      # issued before the host knows: the sticky fault word makes it a no-op ...
        new.step(batches[3])                  # ... and the one after it
    else:
        new.step(batches[2])                  # an unannounced batch: its gather launch targets R[0], the faulted step's input rows
        new.step(batches[3], batches[2])      # ... and a mismatched announcement behind it
    torch.cuda.synchronize()
    assert torch.equal(new.params, p1) and torch.equal(new.adam_m, m1) and torch.equal(new.adam_v, v1)
    st = new.state()              # the state read performs the fall-back: it finishes the faulted step
    assert not st["nan"], st
    assert new.seq_status() == {"enabled": False, "probe": 1, "faults": 1}
    ref.step(batches[1])
    torch.cuda.synchronize()
    assert st["iteration"] == 2 and st == ref.state()
    assert torch.equal(new.params, ref.params) and torch.equal(new.adam_m, ref.adam_m) and torch.equal(new.adam_v, ref.adam_v)
    f = torch.from_numpy(prob["features"][:777]).cuda()
    assert torch.equal(new.get_scene_coordinates(f), ref.get_scene_coordinates(f))   # the 16-bit copies W / W^T of the finished rows too
    for b in batches[4:]:         # the abandoned batches 2 and 3 were not counted; training goes on, on the two-launch flow
        new.step(b); ref.step(b)
    torch.cuda.synchronize()
    assert torch.equal(new.params, ref.params) and torch.equal(new.adam_m, ref.adam_m) and torch.equal(new.adam_v, ref.adam_v)
    assert new.state() == ref.state() and new.state()["iteration"] == 5

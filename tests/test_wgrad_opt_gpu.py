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
from tests.test_chain_gpu import _big_problem
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


def test_expired_exchange_poll_does_not_hang_and_falls_back(diag_lib):
    """Fault injection (ACEZ_WGO_FAULT_AT): in one launch every workgroup waits for a partner count that never comes -- what a tile whose
    two slabs sit on different XCDs looks like. No hang; no weight tile is stored from an incomplete sum (here: none at all); the fault
    word makes every later step a no-op until the next state read, which switches the trainer to the two-launch flow for good."""
    prob = _big_problem(n_images=8, patches_per_view=256)
    os.environ.update(ACEZ_WGO_FAULT_AT="1", ACEZ_SEQ_SPIN_US="3000")
    try:
        new = _small_trainer(prob)
    finally:
        os.environ.pop("ACEZ_WGO_FAULT_AT"); os.environ.pop("ACEZ_SEQ_SPIN_US")
    rng = np.random.default_rng(4)
    batches = [torch.from_numpy(rng.permutation(prob["features"].shape[0])[:2048].astype(np.int64)).cuda() for _ in range(6)]
    new.step(batches[0])
    torch.cuda.synchronize()
    L = new.L
    wide = lambda t: torch.cat([t[l * 262656:l * 262656 + 262144] for l in range(L)])
    w0, m0 = wide(new.params).clone(), wide(new.adam_m).clone()
    new.step(batches[1])          # the faulting step
    torch.cuda.synchronize()
    assert torch.equal(wide(new.params), w0) and torch.equal(wide(new.adam_m), m0)
    p1 = new.params.clone()
    new.step(batches[2])          # issued before the host knows: the sticky fault word makes it a no-op
    torch.cuda.synchronize()
    assert torch.equal(new.params, p1)
    st = new.state()              # the state read performs the fall-back
    assert not st["nan"], st
    assert new.seq_status() == {"enabled": False, "probe": 1, "faults": 1}
    for b in batches[3:]:
        new.step(b)
    torch.cuda.synchronize()
    assert not torch.equal(wide(new.params), w0) and bool(torch.isfinite(new.params).all())
    assert new.state()["iteration"] >= st["iteration"] + 3

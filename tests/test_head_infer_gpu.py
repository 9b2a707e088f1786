"""GPU, diagnostics build (ACEZ_HEAD_INFER=1; a measured alternative of round 5, bit-identical but 15-20 % slower than the large-tile
launches: head_infer.hip): headinfer_kernel -- the head's forward for inference on many rows as ONE launch, a 64-row activation tile
resident in LDS and the weights streamed past it -- against the CPU oracle in bf16 / fp16 mode (1e-3 relative, like every forward
path), against the one-launch chains on small chunks, and against the large-tile launches of the product (bitwise). Covers ragged row counts (the last tile, a wave without valid rows), 0 / 1 / 2 residual blocks (a residual
stream that is written for a later block and one that is not), the non-homogeneous head, and a weight change between two passes
(the packed copy of the weights must follow the optimiser). Reference: ace_network.py:120-149 (Head.forward)."""
import os

import numpy as np
import pytest
import torch

from oracle import head_oracle
from tests import helpers
from tests.test_head_gpu import REL, _rel, _trainer

pytestmark = pytest.mark.gpu


def _features(prob, n, seed, dtype="bf16"):
    rng = np.random.default_rng(seed)
    f = torch.from_numpy(prob["features"][rng.integers(0, len(prob["features"]), size=n)])
    return f.to(torch.float16).to(torch.float32) if dtype == "fp16" else f


@pytest.mark.parametrize("nb,homog,n,dtype", [(1, True, 64, "bf16"), (1, True, 6001, "bf16"), (1, True, 33007, "bf16"), (0, True, 5500, "bf16"),
                                              (2, False, 7777, "bf16"), (1, True, 9000, "fp16"), (2, True, 5121, "fp16")])
def test_one_launch_inference_matches_oracle_and_the_small_chunk_path(nb, homog, n, dtype, diag_lib):
    from acezero_amd.head import HeadTrainer
    from acezero_amd import synth
    prob, _ = helpers.golden_problem()
    flat0 = torch.from_numpy(synth.init_head_params(helpers.SEED + nb, num_head_blocks=nb, use_homogeneous=homog))
    feats = _features(prob, n, 3 + nb, dtype)
    # a pass of more than 5120 rows goes to headinfer_kernel; a smaller one would take the one-launch chains, so for those the chains
    # are switched off (ACEZ_SEQ is one of the two variables the product build reads): then every pass goes to the new kernel
    os.environ["ACEZ_HEAD_INFER"] = "1"
    if n <= 5120:
        os.environ["ACEZ_SEQ"] = "0"
    try:
        big = HeadTrainer(prob["mean"], num_head_blocks=nb, use_homogeneous=homog, max_batch=max(n, 64), iterations=1, dtype=dtype)
    finally:
        os.environ.pop("ACEZ_SEQ", None); os.environ.pop("ACEZ_HEAD_INFER")
    big.load_flat(flat0)
    small = HeadTrainer(prob["mean"], num_head_blocks=nb, use_homogeneous=homog, max_batch=2048, iterations=1, dtype=dtype)        # chunks on the one-launch chains
    small.load_flat(flat0)
    Xb = big.get_scene_coordinates(feats.cuda()).cpu().numpy()
    Xs = small.get_scene_coordinates(feats.cuda()).cpu().numpy()
    assert np.isfinite(Xb).all()
    # same rounding points, different accumulation order inside the matrix instructions (32x32x16 here, 16x16x32 in the chains)
    assert _rel(Xb - prob["mean"], Xs - prob["mean"]) < REL
    orc = head_oracle.HeadOracle(flat0.clone(), prob["mean"], mode=dtype, num_head_blocks=nb, use_homogeneous=homog)
    k = min(n, 3000)
    Xo = orc.scene_coordinates(feats[:k]).numpy()
    assert _rel(Xb[:k] - prob["mean"], Xo - prob["mean"]) < REL
    # the last rows of a ragged pass (the tile whose tail is clamped, the waves that store nothing)
    assert _rel(Xb[-70:] - prob["mean"], Xs[-70:] - prob["mean"]) < REL


def test_one_launch_inference_equals_the_launches_it_replaces(diag_lib):
    """Same MFMA (32x32x16), same K order, same rounding points as the large-tile convolution kernels that ran these passes in round 4
    (ACEZ_HEAD_INFER=0, diagnostics build): the scene coordinates agree to the last bits of the 16-bit activations."""
    from acezero_amd.head import HeadTrainer
    prob, flat0 = helpers.golden_problem()
    n = 33000
    feats = _features(prob, n, 11).cuda()
    os.environ["ACEZ_HEAD_INFER"] = "1"
    try:
        new = HeadTrainer(prob["mean"], max_batch=n, iterations=1)
    finally:
        os.environ.pop("ACEZ_HEAD_INFER")
    new.load_flat(flat0)
    old = HeadTrainer(prob["mean"], max_batch=n, iterations=1)
    old.load_flat(flat0)
    Xn, Xo = new.get_scene_coordinates(feats).cpu().numpy(), old.get_scene_coordinates(feats).cpu().numpy()
    assert np.array_equal(Xn, Xo)


def test_packed_weights_follow_the_optimiser(diag_lib, monkeypatch):
    """The kernel reads a re-packed copy of the 16-bit weights: every path that changes them (fused step, split update, load) must
    invalidate it."""
    prob, flat0 = helpers.golden_problem()
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    monkeypatch.setenv("ACEZ_HEAD_INFER", "1")
    tr = _trainer(prob, flat0, cfg, max_batch=8192, global_batch=512)
    monkeypatch.delenv("ACEZ_HEAD_INFER")
    ref = _trainer(prob, flat0, cfg, max_batch=2048, global_batch=512)
    feats = _features(prob, 8000, 5).cuda()     # 8000 rows: one pass of the new kernel in `tr`, four chunks on the chains in `ref`
    X0 = tr.get_scene_coordinates(feats).cpu().numpy()
    assert _rel(X0 - prob["mean"], ref.get_scene_coordinates(feats).cpu().numpy() - prob["mean"]) < REL
    batches = [torch.from_numpy(b.astype(np.int64)).cuda() for b in helpers.golden_batches(prob, 3)]
    for i, b in enumerate(batches):
        if i == 1:
            tr.backward(b); tr.update()
        else:
            tr.step(b)
        ref.step(b)
    X1 = tr.get_scene_coordinates(feats).cpu().numpy()
    R1 = ref.get_scene_coordinates(feats).cpu().numpy()
    assert _rel(X1 - prob["mean"], R1 - prob["mean"]) < REL and not np.array_equal(X1, X0)
    tr.load_flat(flat0)
    assert _rel(tr.get_scene_coordinates(feats).cpu().numpy() - prob["mean"], X0 - prob["mean"]) < 1e-6

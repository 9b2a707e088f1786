"""GPU: the one-launch chain kernel (head_chain.hip, ACEZ_CHAIN=1) against the per-layer launches (default: rowgemm80 /
loss_kernel) on identical inputs. Same MFMA instruction, same K order and the same rounding points, so
every activation, every propagated gradient, the weight-gradient slabs, the fc3 gradient and the statistics must agree BIT
FOR BIT; only the bias gradients are summed over different row groups (32-row workgroups vs 80-row tiles) and agree to fp32
rounding. Parity with the oracle / the reference goldens is test_head_gpu.py (default path); bit-equality with that path is what
this file adds for the chain kernel."""
import os

import numpy as np
import pytest
import torch

from tests import helpers
from tests.test_head_gpu import _trainer

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _loss_tiles_of_32_rows(monkeypatch, diag_lib):   # diag_lib: chain_kernel and the switches below exist in the diagnostics build only
    """The chain kernel's loss phase sums its fc3 / bias partials over 32-row tiles; the per-layer side of these bitwise comparisons runs
    loss_kernel with the same tile (its default is 16 rows per workgroup: same values to rounding, another summation order)."""
    monkeypatch.setenv("ACEZ_LOSS_ROWS", "8")


def _pair(prob, flat0, cfg, max_batch, num_head_blocks=1):
    out = []
    for chain in ("0", "1"):
        os.environ["ACEZ_CHAIN"] = chain
        os.environ["ACEZ_POSE_TILE"] = "16"   # the chain kernel keeps round 2's pose launches (16-image tiles): same arithmetic on both sides
        try:
            out.append(_trainer(prob, flat0, cfg, max_batch=max_batch))
        finally:
            os.environ.pop("ACEZ_CHAIN", None)
            os.environ.pop("ACEZ_POSE_TILE", None)
    return out


def _big_problem(n_images=24, patches_per_view=512):
    from acezero_amd import synth
    prob = synth.make_training_problem(seed=helpers.SEED + 7, n_images=n_images, views_per_image=2, patches_per_view=patches_per_view)
    prob["features"] = torch.from_numpy(prob["features"]).to(torch.bfloat16).to(torch.float32).numpy()
    return prob


@pytest.mark.parametrize("name,n", [("head_tanh_1cyclepoly", 512), ("head_tanh_1cyclepoly", 5120), ("head_dyntanh_circle", 637),
                                    ("head_tanh_calib", 1000), ("head_tanh_posemlp", 2048), ("head_tanh_posenaive", 333),
                                    ("head_tanh_depth", 512)])
def test_chain_equals_per_layer_launches(name, n):
    prob = _big_problem()
    from oracle import head_oracle
    flat0 = head_oracle.init_params(helpers.SEED + 1)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS[name], prob)
    cfg["global_batch"] = n
    ref, new = _pair(prob, flat0, cfg, max_batch=5120)
    rng = np.random.default_rng(5)
    L = ref.L
    for it in range(3):
        idx = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda()
        new.params.copy_(ref.params); new.adam_m.copy_(ref.adam_m); new.adam_v.copy_(ref.adam_v)
        new.sync_weights()
        if ref.pose_params is not None:
            new.pose_params.copy_(ref.pose_params); new.pose_m.copy_(ref.pose_m); new.pose_v.copy_(ref.pose_v)
        ref.backward(idx)
        new.backward(idx)
        torch.cuda.synchronize()
        # layer inputs of the weight-gradient pass and every dZ: bit for bit
        f1 = L - 2
        stored_out = [l for l in range(L - 1) if not (l < f1 and l % 3 == 2)]   # the chain does not store block outputs (mask only) nor fc2
        for l in stored_out:
            assert np.array_equal(ref.debug_read("out", l, n), new.debug_read("out", l, n)), ("out", l, it)
        for b in range(ref.nb + 2):
            assert np.array_equal(ref.debug_read("R", b, n), new.debug_read("R", b, n)), ("R", b, it)
        for l in range(L):
            assert np.array_equal(ref.debug_read("dZ", l, n), new.debug_read("dZ", l, n)), ("dZ", l, it)
        gr, gn = ref.grad.cpu().numpy(), new.grad.cpu().numpy()
        npar = ref.n_params
        wide = np.arange(L * 262656)
        is_bias = (wide % 262656) >= 262144
        assert np.array_equal(gr[:L * 262656][~is_bias], gn[:L * 262656][~is_bias]), "weight gradients"
        assert np.array_equal(gr[L * 262656:npar + 4], gn[L * 262656:npar + 4]), "fc3 gradient + statistics"
        bias_r, bias_n = gr[:L * 262656][is_bias], gn[:L * 262656][is_bias]
        assert np.array_equal(bias_r[-512:], bias_n[-512:]), "fc2 bias gradient (loss phase: same row groups)"
        np.testing.assert_allclose(bias_n, bias_r, rtol=2e-5, atol=1e-7 * float(np.abs(bias_r).max()))
        if ref.pose_params is not None:
            assert np.array_equal(gr[npar + 4:], gn[npar + 4:]), "pose gradients"
        ref.update()
        new.update()
        torch.cuda.synchronize()
        sr, sn = ref.state(), new.state()
        assert sr["iteration"] == sn["iteration"] == it + 1 and sr["loss"] == sn["loss"] and sr["batch_inliers"] == sn["batch_inliers"]
        np.testing.assert_allclose(new.params.cpu().numpy(), ref.params.cpu().numpy(), rtol=0, atol=2e-6)


def test_chain_fused_step_equals_backward_update():
    """acez_train_step (slabs handed to the optimiser) and backward + update give bitwise equal parameters on the chain path."""
    prob, flat0 = helpers.golden_problem()
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    os.environ["ACEZ_CHAIN"] = "1"
    try:
        a = _trainer(prob, flat0, cfg)
        b = _trainer(prob, flat0, cfg)
    finally:
        os.environ.pop("ACEZ_CHAIN", None)
    for idx in helpers.golden_batches(prob, 6):
        di = torch.from_numpy(idx.astype(np.int64)).cuda()
        a.step(di)
        b.backward(di)
        b.update()
    torch.cuda.synchronize()
    assert torch.equal(a.params, b.params) and torch.equal(a.adam_v, b.adam_v)
    assert a.state() == b.state()


def test_chain_two_head_blocks_and_plain_output():
    """num_head_blocks = 2 (11 wide layers, two residual-gradient fan-ins) and use_homogeneous = False on both paths."""
    from acezero_amd.head import HeadTrainer
    from acezero_amd import synth
    prob = _big_problem(n_images=8, patches_per_view=256)
    n = 1024
    for nb, homog in ((2, True), (1, False), (0, True)):
        trs = []
        for chain in ("0", "1"):
            os.environ["ACEZ_CHAIN"] = chain
            try:
                tr = HeadTrainer(prob["mean"], num_head_blocks=nb, use_homogeneous=homog, max_batch=n, loss_type="tanh", schedule="constant",
                                 iterations=50, lr_min=3e-4)
            finally:
                os.environ.pop("ACEZ_CHAIN", None)
            tr.load_flat(torch.from_numpy(synth.init_head_params(11, num_head_blocks=nb, use_homogeneous=homog)))
            tr.set_buffer(prob["features"], prob["target_px"], prob["view_idx"], prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"],
                          prob["view_image"], prob["image_pose_inv"])
            trs.append(tr)
        ref, new = trs
        rng = np.random.default_rng(3)
        for it in range(3):
            idx = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda()
            ref.step(idx)
            new.step(idx)
        torch.cuda.synchronize()
        assert ref.state()["loss"] == pytest.approx(new.state()["loss"], rel=1e-5)
        np.testing.assert_allclose(new.params.cpu().numpy(), ref.params.cpu().numpy(), rtol=0, atol=1e-5)

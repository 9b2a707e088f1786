"""GPU: the head in the reference's own operand precision. The reference trains and registers under fp16 autocast
(ace_trainer.py:517-518, register_mapping.py:209-210); `HeadTrainer(dtype="fp16")` / ACEZ_DTYPE=fp16 / acez_train_config.compute_dtype
= ACEZ_DTYPE_FP16 run the same kernels with fp16 operands (v_mfma_f32_16x16x32_f16: the bf16 rate), fp32 accumulation, and the gradient
chain scaled by 1024 the way a GradScaler would. Checked here:
  * against the REFERENCE's fp32 goldens (tests/golden/head_*.npz, produced by the reference's TrainerACE.training_step): scene
    coordinates of the first step within 2e-3 relative, first loss within 1e-3 -- bf16 gets 1e-2 / 3-12 % on the same fixtures
    (tests/test_head_gpu.py), because it has three mantissa bits fewer;
  * against the oracle in fp16 mode (same rounding points): 1e-3 on coordinates, the gradient vector, AdamW;
  * one-launch chains vs per-layer launches bit for bit; the fused step vs backward + update bit for bit; no NaN in free-running steps;
  * fp32 (--use_half False) is REJECTED, never silently replaced."""
import os

import numpy as np
import pytest
import torch

from oracle import head_oracle
from tests import helpers
from tests.test_head_gpu import _rel, _trainer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["head_tanh_1cyclepoly", "head_dyntanh_circle", "head_tanh_calib", "head_trained_l1", "head_trained_1cyclepoly",
                                  "head_trained_plain", "head_trained_2blocks"])
def test_fp16_mode_matches_the_reference_fp32_goldens(name):
    prob, flat0, cfg = helpers.problem_for(name)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    tr = _trainer(prob, flat0, cfg, dtype="fp16")
    batches = helpers.golden_batches(prob, cfg["steps"])
    idx0 = batches[0]
    tr.backward(torch.from_numpy(idx0.astype(np.int64)).cuda())
    torch.cuda.synchronize()
    X = tr.last_scene_coords(len(idx0))[:g["coords0"].shape[0]]
    rel = _rel(X - prob["mean"], g["coords0"] - prob["mean"])
    assert rel < helpers.PARITY["scene_coordinates_rel"]["fp16_vs_reference_fp32"], rel                                             # bf16: 1e-2 (tests/test_head_gpu.py: < 3e-2)
    loss0 = float(tr.grad[tr.n_params]) / cfg["global_batch"]
    # first loss: 1e-3 in the untrained regime (residuals of hundreds of pixels); in the trained regime the loss is an L1 norm of 2-6 px
    # residuals and the 11-bit mantissa of fp16 coordinates at metre range is still a few hundredths of a pixel per coordinate: 1.5e-2
    # (bf16: 3 % untrained, up to 12 % trained -- tests/test_head_gpu.py)
    assert abs(loss0 - g["loss"][0]) < (1.5e-2 if name in helpers.TRAINED_CONFIGS else 1e-3) * abs(g["loss"][0]), (loss0, g["loss"][0])
    tr.update()
    # a few free-running steps against the reference's own trajectory (it updates fp32 masters from fp32 gradients; here the
    # gradients went through fp16 activations, so the trajectories separate slowly)
    for idx in batches[1:5]:
        tr.step(torch.from_numpy(idx.astype(np.int64)).cuda())
    st = tr.state()
    assert not st["nan"]
    loss, _ = tr.log(0, min(5, int(g["steps_run"])))
    # (trained regime: sign-gradient losses on few-pixel residuals, the trajectories separate at the rate the bf16 test documents)
    np.testing.assert_allclose(loss, g["loss"][:len(loss)], rtol=0.12 if name in helpers.TRAINED_CONFIGS else 5e-3)


@pytest.mark.parametrize("name", ["head_tanh_1cyclepoly", "head_tanh_calib", "head_tanh_posemlp", "head_tanh_depth", "head_trained_l1sqrt"])
def test_fp16_steps_match_the_fp16_oracle(name):
    prob, flat0, cfg = helpers.problem_for(name)
    tr = _trainer(prob, flat0, cfg, dtype="fp16")
    mlp = cfg["pose_refinement"] in ("mlp", "naive")
    pose_flat = tr.pose_params.cpu().clone() if mlp else None
    orc = head_oracle.TrainerOracle(flat0.clone(), prob["mean"], cfg, mode="fp16", pose_flat=pose_flat, image_pose_inv=prob["image_pose_inv"])
    n_params = flat0.numel()
    for it, idx in enumerate(helpers.golden_batches(prob, 4)):
        b = helpers.torch_batch(prob, idx)
        orc.head.p.flat.copy_(tr.params.cpu())
        orc.sched.m.copy_(tr.adam_m.cpu()); orc.sched.v.copy_(tr.adam_v.cpu())
        if mlp:
            with torch.no_grad():
                orc.pose.flat.copy_(tr.pose_params.cpu())
            orc.pose_m.copy_(tr.pose_m.cpu()); orc.pose_v.copy_(tr.pose_v.cpu())
        rec = orc.step(b["features"], b)
        tr.backward(torch.from_numpy(idx.astype(np.int64)).cuda())
        torch.cuda.synchronize()
        grad = tr.grad.cpu().numpy()
        X = tr.last_scene_coords(len(idx))
        assert _rel(X - prob["mean"], rec["X"].numpy() - prob["mean"]) < 1e-3
        assert abs(grad[n_params] / cfg["global_batch"] - rec["loss"]) < 1e-3 * abs(rec["loss"])
        go = rec["grad"].numpy()
        # (the bounds of the bf16 test: one-ulp flips of activations / gradients between the two summation orders flip relu masks and,
        # in the trained regime, sign(du) -- see tests/test_head_gpu.py)
        assert _rel(grad[:n_params], go) < (5e-2 if name in helpers.TRAINED_CONFIGS else 2e-2), _rel(grad[:n_params], go)
        if mlp:
            assert _rel(grad[n_params + 4:], rec["pose_grad"].numpy()) < 5e-3
        tr.update()
        assert tr.state()["iteration"] == it + 1
        assert _rel(tr.params.cpu().numpy() - flat0.numpy(), orc.head.p.flat.numpy() - flat0.numpy()) < 6e-2


def test_fp16_one_launch_chains_fused_step_and_inference_are_consistent():
    """ACEZ_SEQ=0 / 1 bit for bit, acez_train_step vs backward + update bit for bit, inference (small and multi-chunk) vs the oracle."""
    from tests.helpers import big_problem as _big_problem
    prob = _big_problem(n_images=8, patches_per_view=512)
    flat0 = head_oracle.init_params(helpers.SEED + 1)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    cfg["global_batch"] = 2048
    trs = []
    for seq in ("0", "1", "1"):
        os.environ["ACEZ_SEQ"] = seq
        try:
            trs.append(_trainer(prob, flat0, cfg, max_batch=2048, dtype="fp16"))
        finally:
            os.environ.pop("ACEZ_SEQ", None)
    per_layer, chains, split = trs
    rng = np.random.default_rng(3)
    for it in range(6):
        n = 2048 if it % 2 == 0 else 777
        idx = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda()
        per_layer.step(idx)
        chains.step(idx)
        split.backward(idx)
        split.update()
    torch.cuda.synchronize()
    for other in (chains, split):
        assert torch.equal(per_layer.params, other.params) and torch.equal(per_layer.adam_m, other.adam_m) and torch.equal(per_layer.adam_v, other.adam_v)
    assert per_layer.state() == chains.state() == split.state() and not per_layer.state()["nan"]
    orc = head_oracle.HeadOracle(per_layer.params.cpu().clone(), prob["mean"], mode="fp16")
    for rows in (1, 300, 2048, 5000):
        f = torch.from_numpy(prob["features"][:rows])
        X = chains.get_scene_coordinates(f.cuda()).cpu().numpy()
        assert _rel(X - prob["mean"], orc.scene_coordinates(f).numpy() - prob["mean"]) < 1e-3, rows


def test_fp32_is_rejected_not_replaced():
    from acezero_amd.head import HeadTrainer
    with pytest.raises(NotImplementedError):
        HeadTrainer([0.0, 0.0, 0.0], dtype="fp32")
    os.environ["ACEZ_DTYPE"] = "fp32"
    try:
        with pytest.raises(NotImplementedError):
            HeadTrainer([0.0, 0.0, 0.0])
    finally:
        os.environ.pop("ACEZ_DTYPE")
    # and at the C ABI: compute_dtype = ACEZ_DTYPE_FP32 is an invalid argument
    import ctypes as C
    from acezero_amd import _native as N
    lib = N.lib()
    hd = N.HeadDesc(1, 1, (C.c_float * 3)(0, 0, 0), 0.25, 100.0, 0.9241962407465937)
    cfg = N.TrainConfig()
    cfg.head = hd
    cfg.max_batch = cfg.global_batch = 512
    cfg.iterations = 10
    cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay = 0.9, 0.999, 1e-8, 1e-2
    cfg.compute_dtype = 2
    n = int(lib.acez_head_num_params(C.byref(hd)))
    bufs = [torch.zeros(n + 4, device="cuda") for _ in range(4)]
    pb = N.ParamBuffers(*[C.c_void_p(t.data_ptr()) for t in bufs], n, None, None, None, 0)
    h = C.c_void_p()
    assert lib.acez_trainer_create(C.byref(h), C.byref(cfg), C.byref(pb), 0) == -1
    assert b"fp32" in lib.acez_last_error()

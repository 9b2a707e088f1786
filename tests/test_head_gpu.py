"""GPU: the HIP head path against the CPU oracle in bf16 mode (tolerance 1e-3 relative, BASELINE.json north_star),
against the reference golden vectors (fp32, looser: bf16 vs fp32 arithmetic), and through size-independent
properties at the full BASELINE batch of 5120."""
import os

import numpy as np
import pytest
import torch

from oracle import head_oracle
from tests import helpers

pytestmark = pytest.mark.gpu
REL = helpers.PARITY["scene_coordinates_rel"]["bf16_vs_oracle_bf16"]  # 1e-3: north_star "scene-coordinate tensors within 1e-3 relative fp32" (against the rounding-matched oracle)
REL_REF = helpers.PARITY["scene_coordinates_rel"]   # against the reference's fp32 arithmetic: what 16-bit operands cost


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _trainer(prob, flat0, cfg, max_batch=helpers.B, global_batch=None, dtype=None):
    from acezero_amd.head import HeadTrainer
    tr = HeadTrainer(prob["mean"], dtype=dtype, num_head_blocks=cfg.get("num_head_blocks", 1), use_homogeneous=cfg.get("use_homogeneous", True),
                     max_batch=max_batch, global_batch=global_batch or cfg["global_batch"], loss_type=cfg["loss_type"],
                     schedule=cfg["schedule"], iterations=cfg["iterations"], lr_min=cfg["lr_min"], lr_max=cfg["lr_max"],
                     warmup_iterations=cfg["warmup_iterations"], warmup_lr=cfg["warmup_lr"], cooldown_iterations=cfg["cooldown_iterations"],
                     cooldown_trigger_percent=cfg["cooldown_trigger_percent"], refine_calibration=cfg["refine_calibration"],
                     focal_init=float(prob["focal"]), calib_lr=cfg["calib_lr"], pose_refinement=cfg["pose_refinement"],
                     pose_refinement_wait=cfg["pose_refinement_wait"], pose_seed=helpers.SEED + 3, refinement_ortho=cfg["refinement_ortho"],
                     initial_poses=prob["image_pose_inv"][:, :3] if cfg["pose_refinement"] == "naive" else None)
    tr.load_flat(flat0)
    tr.set_buffer(prob["features"], prob["target_px"], prob["view_idx"], prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"],
                  prob["view_image"], prob["image_pose_inv"], target_crds=prob["target_crds"] if cfg.get("use_depth") else None)
    return tr


def test_inference_scene_coordinates_match_oracle():
    prob, flat0 = helpers.golden_problem()
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    tr = _trainer(prob, flat0, cfg)
    orc = head_oracle.HeadOracle(flat0.clone(), prob["mean"], mode="bf16")
    for n in (512, 77, 1300):   # full tiles, a ragged tail, several chunks of max_batch
        f = torch.from_numpy(prob["features"][:n])
        X = tr.get_scene_coordinates(f.cuda()).cpu().numpy()
        Xo = orc.scene_coordinates(f).numpy()
        assert _rel(X - prob["mean"], Xo - prob["mean"]) < REL
    # vs the reference's own fp32 forward (golden): bf16 arithmetic, so only ~1e-2
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "head_tanh_1cyclepoly.npz"))
    idx0 = helpers.golden_batches(prob, 1)[0][:64]
    X = tr.get_scene_coordinates(torch.from_numpy(prob["features"][idx0]).cuda()).cpu().numpy()
    assert _rel(X - prob["mean"], g["coords0"] - prob["mean"]) < REL_REF["bf16_vs_reference_fp32"]


def _ill_conditioned_rows(tr, orc_dz, n):
    """Rows of the batch whose propagated gradient (dZ of the last wide layer, GPU debug read vs the oracle's) differs by more than 2 % of
    max(its own norm, the median row norm): the criterion of tools/row_gradient_check.py. In the pose-refinement configurations ONE such
    row turns up every dozen steps -- a patch whose loss gradient moves by several percent with the last bits of its refined pose and
    carries up to a fifth of the whole gradient's norm; every other row agrees to bf16 rounding."""
    dzg = torch.from_numpy(tr.debug_read("dZ", tr.L - 1, n).astype(np.int32) << 16).view(torch.float32).numpy()
    diff = np.linalg.norm(dzg - orc_dz, axis=1)
    ref = np.linalg.norm(orc_dz, axis=1)
    off = np.where(diff > 0.02 * np.maximum(ref, np.median(ref) + 1e-30))[0]   # (bf16 rounding alone moves a 512-entry row by ~0.3 % of its norm)
    # ... of which only the HEAVY ones matter for the step's gradient: rows that carry more than 1 % of the propagated gradient's norm (a
    # light row that is off by 5 % moves the whole by < 5e-4; measured on head_tanh_posemlp: rows 410 / 124 with 0.3 % / 0.2 % of the norm in
    # steps 0 / 2, row 491 with 22 % in step 4)
    share = ref[off] / (np.linalg.norm(orc_dz) + 1e-30)
    assert float(np.sqrt((share[share <= 0.01] ** 2).sum())) < 0.01
    return off[share > 0.01]


def _oracle_step_capturing_dz(orc, feats, batch):
    """orc.step with the oracle's per-row dZ of the last wide layer captured on the way (what the GPU's debug read of dZ[L-1] holds)."""
    cap = {}
    inner = orc.head.backward

    def bw(tape, ds):
        f2 = 3 * (orc.head.nb + 1) + 1
        cap["dz"] = orc.head.rg((ds @ orc.head.r(orc.head.p.W3)) * (tape["out"][f2] > 0)).numpy()
        return inner(tape, ds)
    orc.head.backward = bw
    try:
        rec = orc.step(feats, batch)
    finally:
        orc.head.backward = inner
    return rec, cap.get("dz")


@pytest.mark.parametrize("name", list(helpers.HEAD_CONFIGS) + list(helpers.TRAINED_CONFIGS) + ["head_b5120_posemlp_calib"])
def test_training_steps_match_oracle_and_golden(name):
    """Every step compared in isolation (the oracle's weights / optimiser state are re-synchronised with the GPU's before each), all golden
    configurations + the step ace_zero.py runs in its non-seed rounds at BASELINE's batch (pose MLP + focal refinement, 200 images, 5120 rows).
    Per-row criterion (replacing a blanket 2e-2 bound on every step of the pose configurations): rows whose
    propagated gradient is ill-conditioned AND heavy (_ill_conditioned_rows; at most 1 per 512) are taken OUT of the step's batch on both sides and
    the step is compared on the remaining rows at the bounds every other configuration meets: 8e-3 head gradient, 5e-3 pose gradient,
    5e-2 pose update."""
    import copy
    prob, flat0, cfg = helpers.problem_for(name)
    big = name in helpers.BIG_CONFIGS
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    tr = _trainer(prob, flat0, cfg, max_batch=helpers.BIG_B if big else helpers.B)
    mlp = cfg["pose_refinement"] in ("mlp", "naive")
    pose_flat = tr.pose_params.cpu().clone() if mlp else None
    orc = head_oracle.TrainerOracle(flat0.clone(), prob["mean"], cfg, mode="bf16", pose_flat=pose_flat, image_pose_inv=prob["image_pose_inv"])
    batches = helpers.big_batches(prob, cfg["steps"]) if big else helpers.golden_batches(prob, cfg["steps"])
    n_params = flat0.numel()
    removed_total = 0
    for it, idx in enumerate(batches):
        # resynchronise the oracle's weights with the GPU's so that every step is compared in isolation
        orc.head.p.flat.copy_(tr.params.cpu())
        orc.sched.m.copy_(tr.adam_m.cpu()); orc.sched.v.copy_(tr.adam_v.cpu())
        if mlp:
            with torch.no_grad():
                orc.pose.flat.copy_(tr.pose_params.cpu())
            orc.pose_m.copy_(tr.pose_m.cpu()); orc.pose_v.copy_(tr.pose_v.cpu())
            np.testing.assert_allclose(tr.current_poses(), orc.current_poses().numpy(), atol=2e-6)
        start = copy.deepcopy(orc)
        b = helpers.torch_batch(prob, idx)
        di = torch.from_numpy(idx.astype(np.int64)).cuda()
        rec, dz = _oracle_step_capturing_dz(orc, b["features"], b)
        tr.backward(di)
        torch.cuda.synchronize()
        st_before = tr.state()
        if rec is None:
            tr.update()
            assert tr.state()["iteration"] == st_before["iteration"]        # device-side no-op after the schedule ended
            break
        if mlp:
            bad = _ill_conditioned_rows(tr, dz, len(idx))
            assert len(bad) <= (len(idx) + 511) // 512, (it, bad)       # at most one heavy ill-conditioned row per 512
            if len(bad):
                # the step on the batch WITHOUT those rows, on both sides (the loss stays normalised by the global batch)
                removed_total += len(bad)
                idx = np.delete(idx, bad)
                orc = start
                b = helpers.torch_batch(prob, idx)
                di = torch.from_numpy(idx.astype(np.int64)).cuda()
                rec, dz = _oracle_step_capturing_dz(orc, b["features"], b)
                tr.backward(di)
                torch.cuda.synchronize()
                assert len(_ill_conditioned_rows(tr, dz, len(idx))) == 0
        grad = tr.grad.cpu().numpy()
        X = tr.last_scene_coords(len(idx))
        assert _rel(X - prob["mean"], rec["X"].numpy() - prob["mean"]) < REL
        assert abs(grad[n_params] / cfg["global_batch"] - rec["loss"]) < 2e-3 * abs(rec["loss"])
        assert abs(grad[n_params + 1] / cfg["global_batch"] - rec["inliers"]) <= 2.0 / cfg["global_batch"]
        go = rec["grad"].numpy()
        # bf16 roundings of activations / propagated gradients flip by one ulp between the two summation orders: 3-5e-3 on the
        # full gradient vector, depending on the step
        # (trained regime: residuals of a few pixels, and the L1 norm of the 2-vector / the l1 losses have gradient sign(du): a
        # residual that a one-ulp bf16 flip moves across zero flips that patch's whole contribution -- measured up to 3.6e-2)
        assert _rel(grad[:n_params], go) < (5e-2 if name in helpers.TRAINED_CONFIGS else 8e-3), _rel(grad[:n_params], go)
        if name in helpers.TRAINED_CONFIGS:
            cosine = float(np.dot(grad[:n_params].astype(np.float64), go.astype(np.float64)) /
                           (np.linalg.norm(grad[:n_params].astype(np.float64)) * np.linalg.norm(go.astype(np.float64))))
            assert cosine > 0.999, cosine
        if mlp:
            assert _rel(grad[n_params + 4:], rec["pose_grad"].numpy()) < 5e-3, _rel(grad[n_params + 4:], rec["pose_grad"].numpy())
            pose_before = tr.pose_params.cpu().numpy().copy()
        tr.update()
        st = tr.state()
        assert st["iteration"] == it + 1
        assert abs(st["loss"] - rec["loss"]) < 2e-3 * abs(rec["loss"])
        # schedule bookkeeping is exact (the device already holds the cool-down decision of the NEXT iteration)
        orc.sched.check_and_set_cooldown(orc.iteration)
        assert st["max_iterations"] == orc.sched.max_iterations and st["in_cooldown"] == orc.sched.in_cooldown
        assert abs(st["lr"] - orc.sched.lr) <= 1e-15 * max(1.0, abs(orc.sched.lr)) + 1e-18
        if it + 1 < len(g["lr"]):
            assert abs(st["lr"] - g["lr"][it + 1]) < 1e-12
        # AdamW: the GPU applied its own gradient; compare with the oracle's update from the same start
        assert _rel(tr.params.cpu().numpy() - flat0.numpy(), orc.head.p.flat.numpy() - flat0.numpy()) < 6e-2
        if cfg["refine_calibration"]:
            # (the scalar's AdamW state is not re-synchronised between steps: 1e-3 per step, a few percent of that as drift)
            assert abs(st["focal_scale"] - (1.0 + orc.sched.calib_g)) < (6e-5 if name in helpers.TRAINED_CONFIGS else 2e-5)
        if mlp:
            moved = np.abs(tr.pose_params.cpu().numpy() - pose_before).max()
            assert (moved > 0) == (it > cfg["pose_refinement_wait"])          # ace_trainer.py:634: strict >
            if it > cfg["pose_refinement_wait"]:
                assert _rel(tr.pose_params.cpu().numpy() - pose_before, orc.pose.flat.detach().numpy() - pose_before) < 5e-2
            if it < g["poses"].shape[0]:
                # vs the reference PoseRefiner: identical until the first pose update; afterwards each AdamW step moves every
                # weight by ~lr with the sign of a bf16-vs-fp32 gradient, so only the scale of the drift is bounded
                np.testing.assert_allclose(tr.current_poses(), g["poses"][it], atol=1e-5 if it <= cfg["pose_refinement_wait"] else 5e-2)
    assert removed_total <= max(1, len(batches) // 5), removed_total      # (a rare event: one row per dozen steps on the golden problems)
    loss, _ = tr.log(0, min(5, int(g["steps_run"])))
    # vs the reference's own fp32 run: bf16-level agreement. 3 % in the untrained regime; in the trained regime the loss is made of
    # few-pixel reprojection errors and the 8-bit mantissa of the bf16 weights / activations moves it by up to 9 % (the bf16-mode
    # oracle reproduces the GPU's numbers to 2e-3 above; the reference's fp16 autocast has three more mantissa bits)
    # (the 5120-row refinement configuration: the reference's own trajectory follows the signs of its first pose gradients -- its fp32 run
    # and the fp32 oracle are 0.5 % apart on the loss after four pose updates, tests/test_head_oracle.py -- hence 8 % for bf16 there)
    if removed_total == 0:
        np.testing.assert_allclose(loss, g["loss"][:len(loss)], rtol=0.12 if name in helpers.TRAINED_CONFIGS else (8e-2 if big else 3e-2))
    assert tr.state()["max_iterations"] == int(g["max_iterations"][-1])


def test_bf16_weight_copies_track_masters():
    prob, flat0 = helpers.golden_problem()
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_calib"], prob)
    tr = _trainer(prob, flat0, cfg)
    idx = torch.from_numpy(helpers.golden_batches(prob, 1)[0].astype(np.int64)).cuda()
    f = torch.from_numpy(prob["features"][:256]).cuda()
    for _ in range(3):
        tr.step(idx)
    X1 = tr.get_scene_coordinates(f).cpu()
    tr.sync_weights()                                  # recast from the fp32 masters: must be a no-op
    X2 = tr.get_scene_coordinates(f).cpu()
    assert torch.equal(X1, X2)
    sd = tr.state_dict()
    assert set(sd) >= {"res3_conv1.weight", "0c2.bias", "fc3.weight", "mean", "h_beta"} and sd["fc3.weight"].shape == (4, 512, 1, 1)


def test_full_batch_gradient_additivity_and_determinism():
    """BASELINE batch (5120): the gradient of a batch equals the sum of the gradients of its shards when the loss
    normaliser stays the global batch -- the property the data-parallel all-reduce relies on -- and two runs give
    bit-identical gradients (no atomics anywhere)."""
    from acezero_amd import synth
    prob = synth.make_training_problem(seed=7, n_images=20, views_per_image=2, patches_per_view=256)
    flat0 = head_oracle.init_params(11)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    tr = _trainer(prob, flat0, cfg, max_batch=5120, global_batch=5120)
    rng = np.random.default_rng(0)
    idx = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:5120].astype(np.int64)).cuda()
    tr.backward(idx); g_full = tr.grad.clone()
    tr.backward(idx); g_again = tr.grad.clone()
    assert torch.equal(g_full, g_again)
    parts = torch.zeros_like(g_full)
    for lo, hi in ((0, 1900), (1900, 1937), (1937, 5120)):      # ragged shards, like the image-sharded DP split
        tr.backward(idx[lo:hi].contiguous())
        parts += tr.grad
    n = flat0.numel()
    assert _rel(parts[:n].cpu().numpy(), g_full[:n].cpu().numpy()) < 2e-3
    assert abs(float(parts[n] - g_full[n])) < 1e-3 * abs(float(g_full[n]))
    assert float(parts[n + 1]) == float(g_full[n + 1])
    assert torch.isfinite(g_full).all() and float(g_full[:n].abs().sum()) > 0


def test_fused_step_equals_backward_plus_update_bitwise():
    """acez_train_step hands the wgrad slabs straight to AdamW; parameters must match the two-call flow bit for bit."""
    prob, flat0 = helpers.golden_problem()
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    trs = [_trainer(prob, flat0, cfg) for _ in range(2)]
    batches = helpers.golden_batches(prob, 4)
    for idx in batches:
        di = torch.from_numpy(idx.astype(np.int64)).cuda()
        trs[0].step(di)
        trs[1].backward(di)
        trs[1].update()
    torch.cuda.synchronize()
    assert torch.equal(trs[0].params, trs[1].params)
    assert torch.equal(trs[0].adam_m, trs[1].adam_m) and torch.equal(trs[0].adam_v, trs[1].adam_v)
    a, b = trs[0].state(), trs[1].state()
    assert a["iteration"] == b["iteration"] == 4 and a["loss"] == b["loss"] and a["lr"] == b["lr"]


def test_large_batch_inference_runs_on_conv_kernels_and_matches():
    """>= 32768 rows per pass: the head's layers run on the encoder's large-tile implicit-GEMM kernels (1x1 convolutions)."""
    from acezero_amd.head import HeadTrainer
    prob, flat0 = helpers.golden_problem()
    n = 33000                                     # one pass, ragged last 256-row tile
    rng = np.random.default_rng(3)
    feats = torch.from_numpy(prob["features"][rng.integers(0, len(prob["features"]), size=n)])
    big = HeadTrainer(prob["mean"], max_batch=n)
    big.load_flat(flat0)
    small = HeadTrainer(prob["mean"], max_batch=4096)
    small.load_flat(flat0)
    Xb = big.get_scene_coordinates(feats.cuda()).cpu().numpy()
    Xs = small.get_scene_coordinates(feats.cuda()).cpu().numpy()
    # same rounding points, different accumulation order inside the matrix instructions
    assert _rel(Xb - prob["mean"], Xs - prob["mean"]) < REL
    orc = head_oracle.HeadOracle(flat0.clone(), prob["mean"], mode="bf16")
    Xo = orc.scene_coordinates(feats[:4000]).numpy()
    assert _rel(Xb[:4000] - prob["mean"], Xo - prob["mean"]) < REL


def _big_trained(patches_per_view=2048):
    prob, flat0 = helpers.trained_problem(patches_per_view=patches_per_view)   # 6 images x 2 views x 2048 = 24576 patches
    cfg = helpers.full_cfg(helpers.TRAINED_CONFIGS["head_trained_1cyclepoly"], prob)
    cfg.update(global_batch=5120, schedule="constant", lr_min=0.00005, iterations=100)
    return prob, flat0, cfg


@pytest.mark.parametrize("regime", ["untrained", "trained"])
def test_baseline_batch_5120_step_matches_oracle(regime):
    """BASELINE's configuration (batch 5120, default head) against the oracle itself, not only through properties: one step on a
    24 576-patch buffer, in the untrained regime (losses ~35, no inliers) and in the trained one (batch_inliers ~0.86)."""
    if regime == "trained":
        prob, flat0, cfg = _big_trained()
    else:
        from acezero_amd import synth
        prob = synth.make_training_problem(seed=helpers.SEED + 7, n_images=24, views_per_image=2, patches_per_view=512)
        prob["features"] = torch.from_numpy(prob["features"]).to(torch.bfloat16).to(torch.float32).numpy()
        flat0 = head_oracle.init_params(helpers.SEED + 1)
        cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_dyntanh_circle"], prob)
        cfg["global_batch"] = 5120
    tr = _trainer(prob, flat0, cfg, max_batch=5120, global_batch=5120)
    orc = head_oracle.TrainerOracle(flat0.clone(), prob["mean"], cfg, mode="bf16")
    rng = np.random.default_rng(17)
    idx = rng.permutation(prob["features"].shape[0])[:5120]
    b = helpers.torch_batch(prob, idx)
    rec = orc.step(b["features"], b)
    tr.backward(torch.from_numpy(idx.astype(np.int64)).cuda())
    torch.cuda.synchronize()
    n = flat0.numel()
    grad = tr.grad.cpu().numpy()
    X = tr.last_scene_coords(5120)
    assert _rel(X - prob["mean"], rec["X"].numpy() - prob["mean"]) < REL
    assert abs(grad[n] / 5120 - rec["loss"]) < 2e-3 * abs(rec["loss"])
    assert abs(grad[n + 1] / 5120 - rec["inliers"]) <= 3.0 / 5120
    if regime == "trained":
        assert rec["inliers"] > 0.8
    assert _rel(grad[:n], rec["grad"].numpy()) < 8e-3, _rel(grad[:n], rec["grad"].numpy())


@pytest.mark.parametrize("regime", ["untrained", "trained"])
def test_ten_free_running_steps_stay_close_to_the_oracle(regime):
    """No re-synchronisation of weights or optimiser state between steps (test_training_steps_match_oracle_and_golden compares
    every step in isolation): ten consecutive steps of the GPU and of the bf16 oracle from the same start, at BASELINE's batch.
    Stated drift bound (measured, MI355X): per-step loss within 1 % (untrained) / 3 % (trained), batch inliers within 0.5 % of the
    batch; the accumulated parameter movement agrees to 0.3 of its own norm in the untrained regime (AdamW divides by sqrt(v):
    weights whose gradient is at the bf16 noise level still move by +-lr per step, with the sign of that noise) and to < 0.1 in
    the trained regime, with a cosine > 0.95 in both."""
    if regime == "trained":
        prob, flat0, cfg = _big_trained()
    else:
        from acezero_amd import synth
        prob = synth.make_training_problem(seed=helpers.SEED + 7, n_images=24, views_per_image=2, patches_per_view=512)
        prob["features"] = torch.from_numpy(prob["features"]).to(torch.bfloat16).to(torch.float32).numpy()
        flat0 = head_oracle.init_params(helpers.SEED + 1)
        cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
        cfg.update(global_batch=5120, warmup_iterations=1000, iterations=100, cooldown_trigger_percent=0.7)
    tr = _trainer(prob, flat0, cfg, max_batch=5120, global_batch=5120)
    orc = head_oracle.TrainerOracle(flat0.clone(), prob["mean"], cfg, mode="bf16")
    rng = np.random.default_rng(23)
    lo, li = [], []
    for it in range(10):
        idx = rng.permutation(prob["features"].shape[0])[:5120]
        b = helpers.torch_batch(prob, idx)
        rec = orc.step(b["features"], b)
        tr.step(torch.from_numpy(idx.astype(np.int64)).cuda())
        lo.append(rec["loss"]); li.append(rec["inliers"])
    loss, inl = tr.log(0, 10)
    tol = 3e-2 if regime == "trained" else 1e-2
    np.testing.assert_allclose(loss, lo, rtol=tol)
    np.testing.assert_allclose(inl, li, atol=0.005)
    d_gpu = tr.params.cpu().numpy() - flat0.numpy()
    d_orc = orc.head.p.flat.numpy() - flat0.numpy()
    assert _rel(d_gpu, d_orc) < (0.10 if regime == "trained" else 0.40), _rel(d_gpu, d_orc)
    cosine = float(np.dot(d_gpu.astype(np.float64), d_orc.astype(np.float64)) / (np.linalg.norm(d_gpu.astype(np.float64)) * np.linalg.norm(d_orc.astype(np.float64))))
    assert cosine > 0.95, cosine
    assert tr.state()["iteration"] == 10 and abs(tr.state()["lr"] - orc.sched.lr) < 1e-15


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_step_with_the_next_batch_announced_equals_plain_steps_bitwise(dtype):
    """acez_train_step_next gathers the following batch, and closes the step's schedule bookkeeping (in the workgroup that finishes
    last), inside the optimiser's launch. Parameters, optimiser state, schedule state and the per-iteration log must equal plain
    acez_train_step calls bit for bit -- also when the announcement is wrong (another batch follows), when a state read or a split
    step comes in between, with ragged batch sizes, and across the cool-down trigger of 1cyclepoly."""
    from tests.helpers import big_problem as _big_problem
    prob = _big_problem(n_images=8, patches_per_view=512)
    flat0 = head_oracle.init_params(helpers.SEED + 1)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    cfg.update(global_batch=2048, iterations=60, warmup_iterations=5, cooldown_iterations=10, cooldown_trigger_percent=-1.0)
    plain, piped = (_trainer(prob, flat0, cfg, max_batch=2048, dtype=dtype) for _ in range(2))
    rng = np.random.default_rng(17)
    N = prob["features"].shape[0]
    batches = [torch.from_numpy(rng.permutation(N)[:(2048 if i % 5 else 1111)].astype(np.int64)).cuda() for i in range(40)]
    other = torch.from_numpy(rng.permutation(N)[:2048].astype(np.int64)).cuda()
    for i, b in enumerate(batches):
        plain.step(b)
        nxt = batches[i + 1] if i + 1 < len(batches) else None
        if i % 7 == 3:
            nxt = other                       # a wrong announcement: the next call must notice and gather its own batch
        if i % 11 == 5:
            piped.backward(b); piped.update()   # a split step in between (its bookkeeping rides with the next gather as before)
        else:
            piped.step(b, nxt)
        if i % 9 == 4:
            assert plain.state() == piped.state()
    torch.cuda.synchronize()
    assert torch.equal(plain.params, piped.params) and torch.equal(plain.adam_m, piped.adam_m) and torch.equal(plain.adam_v, piped.adam_v)
    sp, sq = plain.state(), piped.state()
    assert sp == sq and sp["iteration"] == sp["max_iterations"] < 60      # the cool-down ended the schedule early, on both
    lp, lq = plain.log(0, sp["iteration"]), piped.log(0, sp["iteration"])
    assert np.array_equal(lp[0], lq[0]) and np.array_equal(lp[1], lq[1])


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_split_step_with_the_next_batch_announced_equals_plain_split_steps_bitwise(dtype):
    """acez_train_update_next (the data-parallel rank's flow: backward / all-reduce / update, with the rank's rows of the next batch
    announced to the update): the optimiser's launch gathers the next batch and closes the step's schedule from the statistics in the
    bucket. Parameters, moments, schedule state, log and the gradient bucket of every step equal backward() + update() bit for bit --
    with wrong announcements, empty announcements, fused steps and state reads in between, ragged sizes, across the cool-down trigger."""
    from tests.helpers import big_problem as _big_problem
    prob = _big_problem(n_images=8, patches_per_view=512)
    flat0 = head_oracle.init_params(helpers.SEED + 1)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    cfg.update(global_batch=2048, iterations=60, warmup_iterations=5, cooldown_iterations=10, cooldown_trigger_percent=-1.0)
    plain, piped = (_trainer(prob, flat0, cfg, max_batch=2048, dtype=dtype) for _ in range(2))
    rng = np.random.default_rng(29)
    N = prob["features"].shape[0]
    batches = [torch.from_numpy(rng.permutation(N)[:(2048 if i % 5 else 1111)].astype(np.int64)).cuda() for i in range(40)]
    other = torch.from_numpy(rng.permutation(N)[:2048].astype(np.int64)).cuda()
    empty = torch.zeros(0, dtype=torch.int64, device="cuda")
    for i, b in enumerate(batches):
        plain.backward(b)
        gp = plain.grad.clone()
        plain.update()
        nxt = batches[i + 1] if i + 1 < len(batches) else None
        if i % 7 == 3:
            nxt = other                       # a wrong announcement: the next backward must notice and gather its own batch
        if i % 13 == 6:
            nxt = empty                       # a rank whose shard holds no row of the next batch
        if i % 11 == 5:
            piped.step(b, nxt)                # a fused step in between
        else:
            piped.backward(b)
            assert torch.equal(gp, piped.grad), i
            piped.update(nxt)
        if i % 9 == 4:
            assert plain.state() == piped.state()
    torch.cuda.synchronize()
    assert torch.equal(plain.params, piped.params) and torch.equal(plain.adam_m, piped.adam_m) and torch.equal(plain.adam_v, piped.adam_v)
    sp, sq = plain.state(), piped.state()
    assert sp == sq and sp["iteration"] == sp["max_iterations"] < 60
    lp, lq = plain.log(0, sp["iteration"]), piped.log(0, sp["iteration"])
    assert np.array_equal(lp[0], lq[0]) and np.array_equal(lp[1], lq[1])


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("name", list(helpers.BIG_CONFIGS))
def test_baseline_batch_against_the_reference_golden(name, dtype):
    """BASELINE's batch (5120 rows) against three steps of the REFERENCE's fp32 training_step (tests/golden/head_b5120_*.npz), in both
    operand types: scene coordinates, the first loss (same weights), then the free-running losses, inlier fractions and focal."""
    prob, flat0, cfg = helpers.problem_for(name)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    tr = _trainer(prob, flat0, cfg, max_batch=helpers.BIG_B, dtype=dtype)
    trained = bool(helpers.BIG_CONFIGS[name].get("trained"))
    batches = helpers.big_batches(prob, cfg["steps"])
    di = [torch.from_numpy(b.astype(np.int64)).cuda() for b in batches]
    tr.backward(di[0])
    torch.cuda.synchronize()
    X = tr.last_scene_coords(helpers.BIG_B)[:64]
    rel = _rel(X - prob["mean"], g["coords0"] - prob["mean"])
    assert rel < (REL_REF["fp16_vs_reference_fp32"] if dtype == "fp16" else REL_REF["bf16_vs_reference_fp32"]), rel
    loss0 = float(tr.grad[tr.n_params]) / helpers.BIG_B
    tol0 = {("fp16", False): 1e-3, ("fp16", True): 1.5e-2, ("bf16", False): 3e-2, ("bf16", True): 0.12}[(dtype, trained)]
    assert abs(loss0 - g["loss"][0]) < tol0 * abs(g["loss"][0]), (loss0, g["loss"][0])
    assert abs(float(tr.grad[tr.n_params + 1]) / helpers.BIG_B - g["inliers"][0]) < (0.03 if trained else 4.0 / helpers.BIG_B)
    tr.update()
    for d in di[1:]:
        tr.step(d)
    st = tr.state()
    assert not st["nan"] and st["iteration"] == cfg["steps"]
    loss, inl = tr.log(0, cfg["steps"])
    refine = cfg["pose_refinement"] == "mlp"
    # (the refinement configuration is a FREE-RUNNING comparison through the pose network's first AdamW steps, which move every weight by
    # +-lr with the sign of its gradient: the reference's own fp32 run and the fp32 oracle separate by 0.5 % on the loss / 1.3e-2 on the poses
    # within four updates (tests/test_head_oracle.py); with 16-bit head arithmetic more signs differ -- measured 7.1 % (bf16) / 4.9 % (fp16)
    # on the loss. The per-step comparison against the reference-pinned oracle is test_training_steps_match_oracle_and_golden.)
    np.testing.assert_allclose(loss, g["loss"], rtol=0.12 if trained else ((8e-2 if dtype == "fp16" else 0.10) if refine else (5e-3 if dtype == "fp16" else 3e-2)))
    # (untrained: 0-2 of 5120 rows are inliers in the reference run; a row at the 10 px threshold may fall on either side)
    np.testing.assert_allclose(inl, g["inliers"], atol=0.03 if trained else (6.0 if refine else 4.0) / helpers.BIG_B)
    if cfg["refine_calibration"]:
        assert abs(st["focal_scale"] - float(g["focal_scale"][-1])) < 2e-3
    if refine:
        # THE step ace_zero.py runs in every non-seed round, as it runs it: acez_train_step_next with the next batch announced (pose forward in
        # the gather launch, per-image reduction + pose backward beside AdamW, pose weight gradients + pose AdamW), 200 images, 5120 rows
        tr2 = _trainer(prob, flat0, cfg, max_batch=helpers.BIG_B, dtype=dtype)
        for i, d in enumerate(di):
            tr2.step(d, di[i + 1] if i + 1 < len(di) else None)
            if i <= cfg["pose_refinement_wait"]:       # no pose update has been applied yet: the reference's refined poses, to fp32 rounding
                np.testing.assert_allclose(tr2.current_poses(), g["poses"][i], atol=1e-5)
        torch.cuda.synchronize()
        loss2, inl2 = tr2.log(0, cfg["steps"])
        assert np.array_equal(loss2, loss) and np.array_equal(inl2, inl)          # the announced flow = the plain flow, bit for bit
        assert torch.equal(tr2.params, tr.params) and torch.equal(tr2.pose_params, tr.pose_params)
        # refined poses of all 200 images after four pose updates: the reference's fp32 run and the fp32 oracle are 1.3e-2 apart here
        # (sign-flipped +-lr weights of the pose network's first AdamW steps); measured with 16-bit head arithmetic: 4.0e-2 / median
        # 4.9e-3 (bf16), 2.3e-2 / 3.8e-3 (fp16) -- the same order as the movement itself (3.9e-2): this early in training the poses'
        # trajectory is sign descent on noise, for the reference too. What IS pinned: the poses before the first update (above, 1e-5)
        # and every single step from a common state (test_training_steps_match_oracle_and_golden).
        dpose = np.abs(tr2.current_poses() - g["poses"][-1])
        assert dpose.max() < 6e-2 and np.median(dpose) < 8e-3, (dpose.max(), np.median(dpose))
        assert abs(tr2.state()["focal_scale"] - float(g["focal_scale"][-1])) < 2e-3


@pytest.mark.parametrize("flow", ["fused", "split", "announced"])
def test_fp16_overflowing_step_is_skipped_and_lowers_the_scale(flow):
    """GradScaler.step (ace_schedule.py:112-113): a step whose propagated gradients left fp16's range applies no head update, does not
    advance AdamW's step count, and the scale drops. The magnitudes are measured before the conversion, so the test drives a FINITE fp32
    value past 65504 (global_batch 1 multiplies every gradient by the batch size; fc3's weights are scaled up until the stored gradient
    of the last wide layer holds an inf): the conversion stores inf while the measured maximum is an ordinary float.
    Round 6 found this window open -- only a literal inf counted as overflow, a 25 000-iteration fp16 refit stepped through it and put
    NaN into the fp32 masters."""
    prob, flat0 = helpers.golden_problem()
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    L = 3 + 3 * cfg.get("num_head_blocks", 1) + 2
    o3 = L * (262144 + 512)
    idx = [torch.from_numpy(np.asarray(b, np.int64)).cuda() for b in helpers.golden_batches(prob, 2)]
    for factor in (1.0, 30.0, 1000.0, 30000.0):
        flat = flat0.clone()
        flat[o3:o3 + 4 * 512] *= factor
        tr = _trainer(prob, flat, cfg, global_batch=1, dtype="fp16")
        s0 = tr.state()
        assert s0["grad_scale"] == 64.0 and s0["opt_steps"] == 0
        before = tr.params.clone()
        if flow == "fused":
            tr.step(idx[0])
        elif flow == "announced":
            tr.step(idx[0], idx[1])
        else:
            tr.backward(idx[0]); tr.update()
        s1 = tr.state()
        dz = tr.debug_read("dZ", L - 1, len(idx[0])).view(np.float16)
        if np.isinf(dz).any():
            break
        assert s1["opt_steps"] == 1 and not torch.equal(tr.params, before)     # in range: an ordinary step
        tr.close()
    else:
        raise AssertionError("no factor drove the gradient out of fp16's range")
    assert not s1["nan"] and s1["iteration"] == 1 and s1["opt_steps"] == 0 and s1["grad_scale"] == 0.25, s1
    assert torch.equal(tr.params, before) and not tr.adam_m.any() and not tr.adam_v.any()
    # the lowered scale brings the same batch back into range (more than one drop when the overflow is by more than 2^8)
    for _ in range(4):
        tr.step(idx[0])
    s2 = tr.state()
    assert s2["iteration"] == 5 and 1 <= s2["opt_steps"] <= 4 and s2["grad_scale"] < 64.0, s2
    assert torch.isfinite(tr.params).all() and torch.isfinite(tr.adam_v).all() and not torch.equal(tr.params, before)

"""A free-running refinement trajectory: 320 consecutive steps of the step ace_zero.py runs in its non-seed rounds
(pose MLP without wait + focal refinement, tanh loss; ace_zero.py:86,97, ace_trainer.py:527,620-640) on a solved mapping problem whose
intrinsics are 3 % too long and whose poses are off by 0.4 degrees / 1.5 cm per image (tests/helpers.py TRAJECTORY).

tests/golden/head_trajectory.npz holds the REFERENCE's own fp32 run (TrainerACE.training_step + PoseRefiner + CalibrationRefiner, made by
tests/golden/make_head_golden.py): loss 13.4 -> 1.9, inlier fraction 0.36 -> 0.96, focal scale 0.999 -> 0.9718 (truth 0.9709), mean pose
error 4.9e-3 -> 3.3e-3. The fp32 oracle (CPU, here) and the HIP trainer in both operand formats (GPU) run the same 320 batches WITHOUT any
re-synchronisation and must follow it: loss curve, inlier fraction, refined poses and focal, with the bounds stated at each assert."""
import os

import numpy as np
import pytest
import torch

from oracle import head_oracle
from tests import helpers

NAME = "head_trajectory"
TRUE_SCALE = 1.0 / helpers.TRAJECTORY["focal_error"]


def _golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", NAME + ".npz"))


def _smooth(x, k=20):
    x = np.asarray(x, np.float64)
    return x[:len(x) // k * k].reshape(-1, k).mean(axis=1)


def trajectory_metrics(loss, inl, focal, poses_last, g, prob):
    """What the asserts below bound, as a dict (tools/ and the tests share it)."""
    true = prob["image_pose_inv_true"][:, :3]
    sl, sg = _smooth(loss), _smooth(g["loss"])
    return {
        "loss_curve_rel_max": float(np.abs(sl / sg - 1).max()), "loss_curve_rel_first100": float(np.abs(sl[:5] / sg[:5] - 1).max()),
        "loss_final_window": float(sl[-1]), "loss_final_window_ref": float(sg[-1]),
        "inliers_window_absmax": float(np.abs(_smooth(inl) - _smooth(g["inliers"])).max()), "inliers_final_window": float(_smooth(inl)[-1]),
        "focal_absmax": float(np.abs(np.asarray(focal) - g["focal_scale"]).max()), "focal_final": float(focal[-1]),
        "pose_vs_ref_max": float(np.abs(poses_last - g["poses"][-1]).max()), "pose_vs_ref_mean": float(np.abs(poses_last - g["poses"][-1]).mean()),
        "pose_err_vs_truth_mean": float(np.abs(poses_last - true).mean()), "pose_err_vs_truth_mean_ref": float(np.abs(g["poses"][-1] - true).mean()),
        "pose_err_vs_truth_mean_start": float(np.abs(prob["image_pose_inv"][:, :3] - true).mean()),
    }


def test_reference_trajectory_converges():
    g = _golden()
    prob, _, cfg = helpers.problem_for(NAME)
    assert int(g["steps_run"]) == cfg["steps"] == 320
    true = prob["image_pose_inv_true"][:, :3]
    assert g["loss"][0] > 13 and _smooth(g["loss"])[-1] < 2.3 and g["inliers"][0] < 0.4 and _smooth(g["inliers"])[-1] > 0.94
    assert abs(g["focal_scale"][-1] - TRUE_SCALE) < 1.5e-3
    assert np.abs(g["poses"][-1] - true).mean() < 0.75 * np.abs(prob["image_pose_inv"][:, :3] - true).mean()   # 4.9e-3 -> 3.3e-3 (via 7.6e-3 after the first updates)


def test_oracle_follows_the_reference_trajectory():
    g = _golden()
    prob, flat0, cfg = helpers.problem_for(NAME)
    from acezero_amd.head import init_pose_network
    tr = head_oracle.TrainerOracle(flat0.clone(), prob["mean"], cfg, mode="fp32", pose_flat=init_pose_network(helpers.SEED + 3),
                                   image_pose_inv=prob["image_pose_inv"])
    loss, inl, focal = [], [], []
    for idx in helpers.golden_batches(prob, cfg["steps"]):
        b = helpers.torch_batch(prob, idx)
        rec = tr.step(b["features"], b)
        loss.append(rec["loss"]); inl.append(rec["inliers"]); focal.append(1.0 + tr.sched.calib_g)
    m = trajectory_metrics(loss, inl, focal, tr.current_poses().numpy(), g, prob)
    np.testing.assert_allclose(loss[:3], g["loss"][:3], rtol=1e-4)          # the same arithmetic at the start
    # two fp32 runs of the same algorithm (different summation orders; AdamW's first steps are sign descent), windows of 20 steps. Measured:
    # loss windows within 2 % over the first 100 steps and at the end, up to 15 % in steps 120-180 (the per-batch loss of the nearly solved
    # problem is a sum of few large residuals: when a handful of rows cross the hard clamp a step apart, a window moves), inlier windows
    # within 0.02, focal within 5e-4 over the whole run, final poses within 1e-3 (mean 2e-4) of the reference's
    assert m["loss_curve_rel_first100"] < 0.04 and m["loss_curve_rel_max"] < 0.25 and abs(m["loss_final_window"] / m["loss_final_window_ref"] - 1) < 0.05, m
    assert m["inliers_window_absmax"] < 0.03 and m["focal_absmax"] < 1e-3, m
    assert m["pose_vs_ref_max"] < 3e-3 and m["pose_vs_ref_mean"] < 5e-4, m
    assert m["pose_err_vs_truth_mean"] < 0.75 * m["pose_err_vs_truth_mean_start"], m


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gpu_follows_the_reference_trajectory(dtype):
    from tests.test_head_gpu import _trainer
    g = _golden()
    prob, flat0, cfg = helpers.problem_for(NAME)
    if dtype == "fp16":
        prob = dict(prob)
        prob["features"] = prob["features"].astype(np.float16).astype(np.float32)     # (bf16-representable values are fp16-representable here: no-op check)
    tr = _trainer(prob, flat0, cfg, dtype=dtype)
    batches = [torch.from_numpy(b.astype(np.int64)).cuda() for b in helpers.golden_batches(prob, cfg["steps"])]
    focal = []
    for i, d in enumerate(batches):
        tr.step(d, batches[i + 1] if i + 1 < len(batches) else None)   # the flow ace_zero runs: next batch announced
        if i % 10 == 9 or i < 3:
            focal.append((i, tr.state()["focal_scale"]))
    torch.cuda.synchronize()
    st = tr.state()
    assert st["iteration"] == cfg["steps"] and not st["nan"]
    loss, inl = tr.log(0, cfg["steps"])
    f_full = np.interp(np.arange(cfg["steps"]), [i for i, _ in focal], [f for _, f in focal])
    m = trajectory_metrics(loss, inl, f_full, tr.current_poses(), g, prob)
    print(dtype, m)
    # 16-bit head arithmetic against the reference's fp32 run, 320 steps without re-synchronisation. Bounds = measured (MI355X, round 6)
    # + a quarter:
    #            loss windows, first 100 steps / whole run   inlier windows   focal (whole run)   final poses max / mean   final loss window
    #   bf16     12.9 % / 36.9 %                             0.034            1.4e-3              4.2e-3 / 5.6e-4         2.47 (reference 2.18)
    #   fp16     16.3 % / 56.8 %                             0.071            1.0e-3              5.9e-3 / 9.8e-4         3.33
    #   (fp32 oracle vs the reference:  1.9 % / 15.4 %       0.019            5e-4                1.0e-3 / 1.9e-4         2.16)
    # fp16 FOLLOWS WORSE THAN bf16 HERE, and the reason is its exponent range, not its mantissa: this fixture's head has small seeded
    # weights (a gradient shrinks 20-fold per layer) next to one unit with weights of ~10^3 (the clamp driver of helpers.trained_problem):
    # the propagated gradients of one step span 1e9 (median 9e-6 in layers 0 / 3 against a maximum of 190 in layers 2 / 5). One power-
    # of-two gradient scale -- the GradScaler scheme of the reference, ace_schedule.py:70,107-113, restated on the device -- puts the
    # maximum near 4096, which leaves a quarter of the first block's gradient entries in fp16's subnormal range and flushes 4 % of them
    # to zero: those layers learn more slowly. bf16 (fp32's exponent) is immune; so is fp32. With PyTorch's default initialisation the
    # per-layer decay is ~2-fold and the effect is absent (tests/test_head_fp16_gpu.py, the B = 5120 goldens above: fp16 10x closer than bf16).
    b = {"bf16": dict(loss=0.45, loss100=0.18, inl=0.045, focal=2.0e-3, pose_max=6e-3, pose_mean=8e-4, inl_final=0.93),
         "fp16": dict(loss=0.70, loss100=0.22, inl=0.090, focal=1.5e-3, pose_max=8e-3, pose_mean=1.3e-3, inl_final=0.86)}[dtype]
    assert m["loss_curve_rel_first100"] < b["loss100"] and m["loss_curve_rel_max"] < b["loss"], m
    assert m["inliers_window_absmax"] < b["inl"] and m["inliers_final_window"] > b["inl_final"], m
    assert m["focal_absmax"] < b["focal"] and abs(m["focal_final"] - TRUE_SCALE) < 2.5e-3, m
    assert m["pose_vs_ref_max"] < b["pose_max"] and m["pose_vs_ref_mean"] < b["pose_mean"], m
    assert m["pose_err_vs_truth_mean"] < 0.8 * m["pose_err_vs_truth_mean_start"], m           # the poses were corrected, as the reference's were

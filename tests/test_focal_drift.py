"""Focal-length refinement over 200 free-running steps (in the session logs the focal moved 525 -> 611).

A solved mapping problem whose intrinsics are 5 % too long (tests/helpers.py FOCAL_DRIFT): the correction 1 + g must go to
1 / 1.05 = 0.9524. tests/golden/head_focal_drift.npz holds the trajectory of the REFERENCE's CalibrationRefiner
(refine_calibration.py:34-59, run by tests/golden/make_head_golden.py); the oracle (CPU, here) and the HIP trainer (GPU) run the same
200 batches without any re-synchronisation and must follow it."""
import os

import numpy as np
import pytest
import torch

from oracle import head_oracle
from tests import helpers

TRUE_SCALE = 1.0 / helpers.FOCAL_DRIFT["focal_error"]


def _golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "head_focal_drift.npz"))


def test_reference_trajectory_converges_to_the_true_focal():
    g = _golden()
    f = g["focal_scale"]
    assert len(f) == 200 and int(g["steps_run"]) == 200
    assert abs(f[-1] - TRUE_SCALE) < 1e-3, f[-1]                       # 0.95269 vs 0.95238
    assert np.all(np.diff(f[:30]) < 0)                                 # AdamW at lr 1e-3: ~ -1e-3 per step at first
    assert np.abs(f[110:] - TRUE_SCALE).max() < 1e-3                   # reached after ~110 steps, no drift away afterwards
    assert g["inliers"][0] < 0.3 and g["inliers"][-1] > 0.9            # 5 % of focal error costs three quarters of the inliers


def test_oracle_follows_the_reference_trajectory():
    g = _golden()
    prob, flat0, cfg = helpers.problem_for("head_focal_drift")
    batches = helpers.golden_batches(prob, cfg["steps"])
    tr = head_oracle.TrainerOracle(flat0.clone(), prob["mean"], cfg, mode="fp32", image_pose_inv=prob["image_pose_inv"])
    focal, loss = [], []
    for idx in batches:
        b = helpers.torch_batch(prob, idx)
        rec = tr.step(b["features"], b)
        focal.append(1.0 + tr.sched.calib_g)
        loss.append(rec["loss"])
    np.testing.assert_allclose(focal, g["focal_scale"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(loss[:5], g["loss"][:5], rtol=3e-4)
    # free-running: once the problem is nearly solved (loss 14 -> 2) the per-batch loss is a sum of few large residuals and the two
    # runs' weights differ in the last bits (AdamW on gradients at the rounding level); the focal correction is insensitive to that
    rel = np.abs(np.array(loss) - g["loss"]) / g["loss"]
    assert np.median(rel) < 0.03 and rel[:40].max() < 0.02 and rel.max() < 0.35, (np.median(rel), rel[:40].max(), rel.max())


@pytest.mark.gpu
def test_gpu_follows_the_reference_trajectory():
    from tests.test_head_gpu import _trainer
    g = _golden()
    prob, flat0, cfg = helpers.problem_for("head_focal_drift")
    batches = helpers.golden_batches(prob, cfg["steps"])
    tr = _trainer(prob, flat0, cfg)
    focal = []
    for idx in batches:
        tr.step(torch.from_numpy(idx.astype(np.int64)).cuda())
        focal.append(tr.state()["focal_scale"])
    focal = np.array(focal)
    # bf16 kernels against the reference's fp32 arithmetic, 200 steps without re-synchronisation
    assert np.abs(focal - g["focal_scale"]).max() < 2e-3, np.abs(focal - g["focal_scale"]).max()
    assert abs(focal[-1] - TRUE_SCALE) < 1.5e-3, focal[-1]
    loss, _ = tr.log(0, 200)
    rel = np.abs(np.asarray(loss) - g["loss"]) / g["loss"]
    # while the focal error dominates the loss (first 40 steps, 14 -> 6 px) the two agree to a few percent (median 1 %, single batches up to 8 %); on the solved problem
    # (2 px) the bf16 rounding of the predicted coordinates (8 significant bits: millimetres at metre range = tenths of a pixel per
    # coordinate) is a visible part of the L1 residual: the GPU's loss sits 20-30 % above the fp32 reference's there
    assert rel[:40].max() < 0.10 and np.median(rel[:40]) < 0.03 and np.median(rel) < 0.15 and rel.max() < 0.5, (rel[:40].max(), np.median(rel), rel.max())

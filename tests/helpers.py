"""Shared helpers of the parity tests (inputs identical to tests/golden/make_head_golden.py)."""
import os

import numpy as np
import torch

from acezero_amd import synth
from oracle import head_oracle

B = 512
SEED = 2089

HEAD_CONFIGS = {
    "head_tanh_1cyclepoly": dict(loss_type="tanh", schedule="1cyclepoly", lr_min=0.0001, lr_max=0.0006, warmup_iterations=4,
                                 warmup_lr=0.0001, cooldown_iterations=5, cooldown_trigger_percent=-1.0, iterations=40,
                                 refine_calibration=False, steps=16),
    "head_dyntanh_circle": dict(loss_type="dyntanh", schedule="circle", lr_min=0.0001, lr_max=0.001, warmup_iterations=1000,
                                warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                                refine_calibration=False, steps=10),
    "head_tanh_calib": dict(loss_type="tanh", schedule="constant", lr_min=0.0002, lr_max=0.003, warmup_iterations=1000,
                            warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                            refine_calibration=True, steps=8),
    "head_tanh_posemlp": dict(loss_type="tanh", schedule="constant", lr_min=0.0002, lr_max=0.003, warmup_iterations=1000,
                              warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                              refine_calibration=False, steps=6, pose_refinement="mlp", pose_refinement_wait=2),
    "head_tanh_posenaive": dict(loss_type="tanh", schedule="constant", lr_min=0.0002, lr_max=0.003, warmup_iterations=1000,
                                warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                                refine_calibration=False, steps=5, pose_refinement="naive", pose_refinement_wait=0),
    "head_tanh_depth": dict(loss_type="tanh", schedule="constant", lr_min=0.0002, lr_max=0.003, warmup_iterations=1000,
                            warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                            refine_calibration=False, steps=6, use_depth=True),
    "head_tanh_posemlp_procrustes": dict(loss_type="tanh", schedule="constant", lr_min=0.0002, lr_max=0.003, warmup_iterations=1000,
                                         warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                                         refine_calibration=False, steps=6, pose_refinement="mlp", pose_refinement_wait=2,
                                         refinement_ortho="procrustes"),
}


def full_cfg(c, prob=None):
    d = dict(c)
    d.update(global_batch=B, soft_clamp=50.0, soft_clamp_min=1.0, circle_schedule=True, hard_clamp=1000.0,
             depth_min=0.1, depth_max=1000.0, depth_target=10.0, inlier_px_threshold=10.0, num_head_blocks=1,
             use_homogeneous=True, calib_lr=0.001)
    d.setdefault("pose_refinement", "none")
    d.setdefault("pose_refinement_wait", 0)
    d.setdefault("use_depth", False)
    d.setdefault("refinement_ortho", "gram-schmidt")
    if prob is not None:
        d["focal_init"] = float(prob["focal"])
    return d


def golden_problem():
    prob = synth.make_training_problem(seed=SEED, n_images=6, views_per_image=2, patches_per_view=128)
    prob["features"] = torch.from_numpy(prob["features"]).to(torch.bfloat16).to(torch.float32).numpy()
    flat0 = head_oracle.init_params(SEED + 1)
    return prob, flat0


# ---- the "trained regime" problem (every round-1 golden lived at batch_inliers <= 0.004) -------------------------
# Configurations that run on it: the reprojection-loss variants, two head blocks, the non-homogeneous head, and an early stop
# whose cool-down trigger (min of the last batch_inliers > 0.7, ace_schedule.py:86-101) fires on real values.
TRAINED_CONFIGS = {
    "head_trained_1cyclepoly": dict(loss_type="tanh", schedule="1cyclepoly", lr_min=0.00002, lr_max=0.0001, warmup_iterations=3,
                                    warmup_lr=0.00002, cooldown_iterations=6, cooldown_trigger_percent=0.7, iterations=40,
                                    refine_calibration=False, steps=14),
    "head_trained_l1": dict(loss_type="l1", schedule="constant", lr_min=0.00005, lr_max=0.003, warmup_iterations=1000, warmup_lr=0.0005,
                            cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20, refine_calibration=False, steps=6),
    "head_trained_l1sqrt": dict(loss_type="l1+sqrt", schedule="constant", lr_min=0.00005, lr_max=0.003, warmup_iterations=1000,
                                warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                                refine_calibration=False, steps=6),
    "head_trained_l1log": dict(loss_type="l1+log", schedule="constant", lr_min=0.00005, lr_max=0.003, warmup_iterations=1000,
                               warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                               refine_calibration=False, steps=6),   # train_ace.py:158 spells it 'l1+log': ace_loss.py's else branch
    "head_trained_2blocks": dict(loss_type="dyntanh", schedule="constant", lr_min=0.00005, lr_max=0.003, warmup_iterations=1000,
                                 warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                                 refine_calibration=False, steps=6, num_head_blocks=2),
    "head_trained_plain": dict(loss_type="tanh", schedule="constant", lr_min=0.00005, lr_max=0.003, warmup_iterations=1000,
                               warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                               refine_calibration=True, steps=6, use_homogeneous=False),
}


# ---- focal-length refinement over 200 free-running steps (the session logs showed the focal drifting 525 -> 611) ----
# The trained-regime problem, but the intrinsics the trainer is given are 5 % too long (view_K and focal_init, as a wrong
# --use_external_focal_length would make them; the poses, the targets and the solved head belong to the true focal). The true
# correction is known: focal_scale = 1 + g -> 1 / 1.05 = 0.9524. Reference (CalibrationRefiner, refine_calibration.py:34-59), oracle and
# kernels are compared on the whole trajectory of 1 + g.
FOCAL_DRIFT = dict(loss_type="tanh", schedule="constant", lr_min=0.00002, lr_max=0.0002, warmup_iterations=1000, warmup_lr=0.0005,
                   cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=400, refine_calibration=True, steps=200,
                   focal_error=1.05)


# ---- a free-running refinement trajectory: 320 steps of the step ace_zero.py runs in its non-seed rounds -- pose MLP
# (no wait) + focal refinement, tanh loss -- on the solved problem with a 3 % focal error AND perturbed poses (0.4 degrees / 1.5 cm per
# image): the loss falls as the focal and the poses are corrected; loss curve, inlier fraction, refined poses and focal of the REFERENCE's
# fp32 run are the fixture (tests/golden/head_trajectory.npz), the oracle (CPU) and the HIP trainer in both operand formats follow it.
TRAJECTORY = dict(loss_type="tanh", schedule="constant", lr_min=0.00002, lr_max=0.0002, warmup_iterations=1000, warmup_lr=0.0005,
                  cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=1000, refine_calibration=True, steps=320,
                  focal_error=1.03, pose_refinement="mlp", pose_refinement_wait=0, pose_noise=(0.4, 0.015))


def perturb_poses(image_pose_inv, deg, metres, seed):
    """world -> camera matrices [I,4,4] left-multiplied by a small seeded rigid motion each (float64 arithmetic, float32 result)."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    out = image_pose_inv.astype(np.float64).copy()
    for i in range(out.shape[0]):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        D = np.eye(4)
        D[:3, :3] = Rotation.from_rotvec(ax * np.deg2rad(deg)).as_matrix()
        t = rng.normal(size=3)
        D[:3, 3] = t / np.linalg.norm(t) * metres
        out[i] = D @ out[i]
    return out.astype(np.float32)


def trained_problem(num_head_blocks=1, use_homogeneous=True, patches_per_view=128):
    """A training problem and head weights that ALREADY solve it, built without any training so that the reference, the oracle and
    the kernels can start from bit-identical numbers on any machine: the synthetic features are (almost) linear in the scene
    coordinate (small feature_gain), the residual blocks carry small seeded weights, fc1 holds the closed-form linear read-out
    M = proj^T (proj proj^T)^-1 / gain as +M / -M ReLU pairs, fc2 passes them on and fc3 recombines them (relu(z) - relu(-z) = z).
    ~6 % of the rows have garbage features (outliers: e > 10 px, the large-error branches of the l1 variants, invalid depths), and
    with the homogeneous output a seventh fc1 unit drives s3 past the h clamp (ace_network.py:143) for the top few percent of a
    random feature direction, so clamp(max=min_inv_scale) fires with real values. Returns (prob, flat0) like golden_problem()."""
    gain, noise = 0.03, 0.002
    prob = synth.make_training_problem(seed=SEED + 11, n_images=6, views_per_image=2, patches_per_view=patches_per_view, feature_noise=noise,
                                       feature_gain=gain)
    rng = np.random.default_rng(SEED + 12)
    feats = prob["features"].astype(np.float64)
    n = feats.shape[0]
    bad = rng.uniform(size=n) < 0.06
    feats[bad] = rng.normal(0.0, 0.05, size=(int(bad.sum()), 512))
    prob["features"] = torch.from_numpy(feats.astype(np.float32)).to(torch.bfloat16).to(torch.float32).numpy()
    no = 4 if use_homogeneous else 3
    flat = head_oracle.init_params(SEED + 13, num_head_blocks, use_homogeneous, scale=0.05).double().numpy()
    L = 3 + 3 * num_head_blocks + 2
    f1, f2 = L - 2, L - 1
    W = lambda l: flat[l * 262656: l * 262656 + 262144].reshape(512, 512)
    b = lambda l: flat[l * 262656 + 262144: (l + 1) * 262656]
    W3 = flat[L * 262656: L * 262656 + no * 512].reshape(no, 512)
    b3 = flat[L * 262656 + no * 512:]
    proj = prob["feature_proj"].astype(np.float64)
    M = proj.T @ np.linalg.inv(proj @ proj.T) / gain          # [512, 3]
    W(f1)[0:3] = M.T
    W(f1)[3:6] = -M.T
    b(f1)[0:7] = 0.0
    W(f1)[6] = 0.0
    W(f2)[0:7] = 0.0
    W(f2)[np.arange(7), np.arange(7)] = 1.0
    b(f2)[0:7] = 0.0
    W3[:] = 0.0
    b3[:] = 0.0
    for j in range(3):
        W3[j, j], W3[j, j + 3] = 1.0, -1.0
    if use_homogeneous:
        u = rng.normal(size=512)
        u /= np.linalg.norm(u)
        z = prob["features"].astype(np.float64) @ u
        q93, q97 = np.quantile(z, 0.93), np.quantile(z, 0.97)
        c = 140.0 / (q97 - q93)
        W(f1)[6] = c * u
        b(f1)[6] = -c * q93
        W3[3, 6] = 1.0
    return prob, torch.from_numpy(flat.astype(np.float32))


# ---- BASELINE's batch (VERDICT r2 weakness 10: every reference golden lived at B = 512; 5120 was compared with the oracle only) ----
# 6 images x 2 views x 512 patches = 6144 rows, batches of 5120, three steps of the reference itself; one untrained, one trained problem.
BIG_B = 5120
BIG_CONFIGS = {
    "head_b5120_tanh": dict(loss_type="tanh", schedule="1cyclepoly", lr_min=0.0001, lr_max=0.0006, warmup_iterations=2, warmup_lr=0.0001,
                            cooldown_iterations=5, cooldown_trigger_percent=-1.0, iterations=40, refine_calibration=False, steps=3),
    "head_b5120_trained": dict(loss_type="dyntanh", schedule="constant", lr_min=0.00005, lr_max=0.003, warmup_iterations=1000, warmup_lr=0.0005,
                               cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20, refine_calibration=True, steps=3,
                               trained=True),
    # THE step every non-seed round of ace_zero.py runs (ace_zero.py:86,97: --pose_refinement mlp --refine_calibration True; tanh, 1cyclepoly
    # with ace_zero's learning rates), at BASELINE's batch on 200 images: 6 steps, the pose network moving from the third on (wait = 1)
    "head_b5120_posemlp_calib": dict(loss_type="tanh", schedule="1cyclepoly", lr_min=0.0005, lr_max=0.003, warmup_iterations=1000, warmup_lr=0.0005,
                                     cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=25000, refine_calibration=True, steps=6,
                                     pose_refinement="mlp", pose_refinement_wait=1, n_images=200),
}


def big_problem_for(name):
    c = BIG_CONFIGS[name]
    if c.get("trained"):
        prob, flat0 = trained_problem(patches_per_view=512)
    elif "n_images" in c:   # many images, few patches each (200 x 2 views x 16 = 6400 rows): every image's pose takes part in every batch
        prob = synth.make_training_problem(seed=SEED + 31, n_images=c["n_images"], views_per_image=2, patches_per_view=16)
        prob["features"] = torch.from_numpy(prob["features"]).to(torch.bfloat16).to(torch.float32).numpy()
        flat0 = head_oracle.init_params(SEED + 1)
    else:
        prob = synth.make_training_problem(seed=SEED + 21, n_images=6, views_per_image=2, patches_per_view=512)
        prob["features"] = torch.from_numpy(prob["features"]).to(torch.bfloat16).to(torch.float32).numpy()
        flat0 = head_oracle.init_params(SEED + 1)
    cfg = full_cfg({k: v for k, v in c.items() if k not in ("trained", "n_images")}, prob)
    cfg["global_batch"] = BIG_B
    return prob, flat0, cfg


def big_batches(prob, steps):
    rng = np.random.default_rng(SEED + 22)
    n = prob["features"].shape[0]
    return [rng.permutation(n)[:BIG_B] for _ in range(steps)]


def problem_for(name):
    """(prob, flat0, cfg) of a golden configuration."""
    if name in BIG_CONFIGS:
        return big_problem_for(name)
    if name in TRAINED_CONFIGS or name in ("head_focal_drift", "head_trajectory"):
        c = TRAINED_CONFIGS[name] if name in TRAINED_CONFIGS else (FOCAL_DRIFT if name == "head_focal_drift" else TRAJECTORY)
        prob, flat0 = trained_problem(c.get("num_head_blocks", 1), c.get("use_homogeneous", True))
        if "focal_error" in c:
            prob["view_K"] = prob["view_K"].copy()
            prob["view_K"][:, 0, 0] *= np.float32(c["focal_error"])
            prob["view_K"][:, 1, 1] *= np.float32(c["focal_error"])
            prob["view_Kinv"] = np.linalg.inv(prob["view_K"].astype(np.float64)).astype(np.float32)
            prob["focal"] = np.float32(float(prob["focal"]) * c["focal_error"])
        if "pose_noise" in c:
            prob["image_pose_inv_true"] = prob["image_pose_inv"].copy()
            prob["image_pose_inv"] = perturb_poses(prob["image_pose_inv"], c["pose_noise"][0], c["pose_noise"][1], SEED + 41)
        cfg = full_cfg({k: v for k, v in c.items() if k != "pose_noise"}, prob)
        cfg["num_head_blocks"] = c.get("num_head_blocks", 1)
        cfg["use_homogeneous"] = c.get("use_homogeneous", True)
        return prob, flat0, cfg
    prob, flat0 = golden_problem()
    return prob, flat0, full_cfg(HEAD_CONFIGS[name], prob)


def golden_batches(prob, steps):
    rng = np.random.default_rng(SEED + 2)
    n = prob["features"].shape[0]
    return [rng.permutation(n)[:B] for _ in range(steps)]


def torch_batch(prob, idx):
    pp = synth.expand_per_patch(prob, idx)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in pp.items()}


# ---- point-cloud extraction cases (tests/golden/cloud_cases.npz holds the reference's output for each) ----
# name -> (synth seed, frames, map h, w, noise sigma [m], outlier ratio, len(data_loader), filter_depth, dense)
CLOUD_CASES = {
    "plain": (11, 3, 24, 32, 0.0004, 0.1, 1000, 100.0, False),
    "relaxed": (12, 3, 24, 32, 0.02, 0.2, 500, 100.0, False),
    "subsampled": (13, 3, 24, 32, 0.0004, 0.05, 10000, 100.0, False),
    "depth_all_out": (14, 2, 24, 32, 0.0004, 0.1, 1000, 0.01, False),
    "depth_some_out": (15, 3, 24, 32, 0.0004, 0.1, 1000, 2.5, False),
    "grad_escalation_inf": (16, 3, 24, 32, 0.001, 0.55, 200, 100.0, False),
    "grad_escalation_mid": (20, 3, 24, 32, 0.06, 0.1, 250, 100.0, False),
    "dense": (17, 2, 24, 32, 0.02, 0.3, 1000, 3.0, True),
    "dense_relaxed": (21, 2, 24, 32, 0.02, 0.3, 100, 100.0, True),
    "dense_subsampled": (18, 2, 24, 32, 0.02, 0.3, 20000, 3.0, True),
    "wide_map": (19, 2, 15, 93, 0.0004, 0.1, 500, 100.0, False),
}
CLOUD_RANDOM_CASES = ("subsampled", "dense_subsampled")   # torch.randperm decides there: only count / subset are comparable


def cloud_case_inputs(name):
    """(scene coords [n,3,h,w], poses_inv [n,4,4] world->camera, K [n,3,3], len(data_loader), filter_depth, dense)."""
    import numpy as np
    from acezero_amd import synth
    seed, n, h, w, sigma, outl, loader_len, depth, dense = CLOUD_CASES[name]
    fr = synth.make_registration_frames(seed=seed, n_frames=n, h=h, w=w, noise_sigma=sigma, outlier_ratio=outl)
    poses_inv = np.linalg.inv(fr["poses"]).astype(np.float32)
    K = np.array([[fr["focal"], 0, fr["ppx"]], [0, fr["focal"], fr["ppy"]], [0, 0, 1]], np.float32)
    return fr["scene_coords"], poses_inv, np.stack([K] * n), loader_len, depth, dense


# ---- pose-file cases (tests/golden/pose_file_ref.txt / .npz hold what the reference's writer / reader make of them) ----
def pose_file_cases(n=12, seed=5):
    """world -> camera matrices (float64) and confidences, seeded."""
    import numpy as np
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    out = np.tile(np.eye(4), (n, 1, 1))
    out[:, :3, :3] = Rotation.from_rotvec(rng.normal(0, 1.0, size=(n, 3))).as_matrix()
    out[:, :3, 3] = rng.normal(0, 3.0, size=(n, 3))
    conf = rng.integers(0, 3000, size=n)
    conf[1], conf[2] = 499, 500          # the threshold itself is kept (confidence < threshold is dropped), 499 is not
    return out, conf


def _parity_tolerances():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_tolerances.json")) as f:
        return json.load(f)


# the tolerance table of the T path: one file read by the tests that assert it and by bench.py, which quotes it on its JSON line
PARITY = _parity_tolerances()


def big_problem(n_images=24, patches_per_view=512):
    """24 images x 2 views x 512 patches with bf16-representable features: the problem of the bitwise flow-equivalence tests."""
    prob = synth.make_training_problem(seed=SEED + 7, n_images=n_images, views_per_image=2, patches_per_view=patches_per_view)
    prob["features"] = torch.from_numpy(prob["features"]).to(torch.bfloat16).to(torch.float32).numpy()
    return prob

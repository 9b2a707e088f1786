"""Generate tests/golden/head_*.npz by running the REFERENCE's own code on CPU (fp32).

Run in the build container only (needs /root/reference):   python tests/golden/make_head_golden.py

What runs is the unmodified ace_trainer.TrainerACE.training_step (ace_trainer.py:499-679) with the reference's
ace_network.Head, ace_loss.ReproLoss, ace_schedule.ScheduleACE, refine_calibration.CalibrationRefiner, on a
TrainerACE instance created with object.__new__ and hand-set attributes (the constructor needs a dataset on
disk and a CUDA device).  Modules the container lacks and this path never calls (torchvision, skimage, roma,
pyrender, trimesh, matplotlib.pyplot) are stubbed.  Inputs and initial weights come from seeded numpy
generators (acezero_amd.synth, oracle.head_oracle.init_params) so the fixtures only hold OUTPUTS.
"""
import io
import os
import sys
import time
import types
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"

for name in ["torchvision", "torchvision.transforms", "torchvision.transforms.functional", "skimage", "skimage.transform",
             "skimage.io", "skimage.color", "skimage.draw", "pyrender", "trimesh", "matplotlib.pyplot",
             "ace_visualizer", "ace_vis_util"]:
    sys.modules.setdefault(name, MagicMock())
# roma 1.4.1 (environment.yml:226) is not vendored and not installed: the one function the path calls
# (refine_poses.py:148) is restated in oracle.head_oracle.special_gramschmidt -- parity unpinned for that call
import types as _types  # noqa: E402
_roma = _types.ModuleType("roma")
sys.modules["roma"] = _roma
sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self  # refine_calibration.py:42 calls .cuda()

import ace_loss  # noqa: E402
import ace_network  # noqa: E402
import ace_schedule  # noqa: E402
import ace_trainer  # noqa: E402
import refine_calibration  # noqa: E402
from acezero_amd import synth  # noqa: E402
from oracle import head_oracle  # noqa: E402
import refine_poses  # noqa: E402
_roma.special_gramschmidt = head_oracle.special_gramschmidt
_roma.special_procrustes = head_oracle.special_procrustes

CONFIGS = {
    # ace_zero's mapping settings (ace_zero.py:105-123): tanh, 1cyclepoly, lr_max 0.003
    "head_tanh_1cyclepoly": dict(loss_type="tanh", schedule="1cyclepoly", lr_min=0.0001, lr_max=0.0006, warmup_iterations=4,
                                 warmup_lr=0.0001, cooldown_iterations=5, cooldown_trigger_percent=-1.0, iterations=40,
                                 refine_calibration=False, steps=16),
    # train_ace.py defaults: dyntanh + OneCycle "circle", lr_max 0.005
    "head_dyntanh_circle": dict(loss_type="dyntanh", schedule="circle", lr_min=0.0001, lr_max=0.001, warmup_iterations=1000,
                                warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                                refine_calibration=False, steps=10),
    # focal-length refinement in the loop (BASELINE config 3)
    "head_tanh_calib": dict(loss_type="tanh", schedule="constant", lr_min=0.0002, lr_max=0.003, warmup_iterations=1000,
                            warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                            refine_calibration=True, steps=8),
    # ace_zero's pose refinement (ace_zero.py:86): PoseNetwork(0,128) + Gram-Schmidt, own AdamW after a wait of 2 iterations
    "head_tanh_posemlp": dict(loss_type="tanh", schedule="constant", lr_min=0.0002, lr_max=0.003, warmup_iterations=1000,
                              warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                              refine_calibration=False, steps=6, pose_refinement="mlp", pose_refinement_wait=2),
    # depth-supervised mapping (the seed stage of ace_zero: use_depth, ace_trainer.py:567-574,601-609)
    "head_tanh_posenaive": dict(loss_type="tanh", schedule="constant", lr_min=0.0002, lr_max=0.003, warmup_iterations=1000,
                                warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                                refine_calibration=False, steps=5, pose_refinement="naive", pose_refinement_wait=0),
    "head_tanh_depth": dict(loss_type="tanh", schedule="constant", lr_min=0.0002, lr_max=0.003, warmup_iterations=1000,
                            warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                            refine_calibration=False, steps=6, use_depth=True),
    # --refinement_ortho procrustes (train_ace.py:226): nearest-rotation orthonormalisation of the updated poses
    "head_tanh_posemlp_procrustes": dict(loss_type="tanh", schedule="constant", lr_min=0.0002, lr_max=0.003, warmup_iterations=1000,
                                         warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, iterations=20,
                                         refine_calibration=False, steps=6, pose_refinement="mlp", pose_refinement_wait=2,
                                         refinement_ortho="procrustes"),
}
B = 512
SEED = 2089


def full_cfg(c):
    d = dict(c)
    d.update(global_batch=B, soft_clamp=50.0, soft_clamp_min=1.0, circle_schedule=True, hard_clamp=1000.0,
             depth_min=0.1, depth_max=1000.0, depth_target=10.0, inlier_px_threshold=10.0, num_head_blocks=1,
             use_homogeneous=True, calib_lr=0.001)
    d.setdefault("pose_refinement", "none")
    d.setdefault("pose_refinement_wait", 0)
    d.setdefault("use_depth", False)
    d.setdefault("refinement_ortho", "gram-schmidt")
    return d


def run_reference(cfg, prob, flat0, batches):
    nb, homog = cfg.get("num_head_blocks", 1), cfg.get("use_homogeneous", True)
    opt = types.SimpleNamespace(
        use_half=False, depth_min=cfg["depth_min"], depth_max=cfg["depth_max"], depth_target=cfg["depth_target"],
        repro_loss_hard_clamp=cfg["hard_clamp"], learning_rate_cooldown_trigger_px_threshold=cfg["inlier_px_threshold"],
        pose_refinement_wait=cfg["pose_refinement_wait"], pose_refinement=cfg["pose_refinement"], pose_refinement_lr=0.001,
        pose_refinement_weight=0.1, refinement_ortho=cfg["refinement_ortho"], iterations=cfg["iterations"], learning_rate_schedule=cfg["schedule"],
        learning_rate_min=cfg["lr_min"], learning_rate_max=cfg["lr_max"], learning_rate_warmup_iterations=cfg["warmup_iterations"],
        learning_rate_warmup_learning_rate=cfg["warmup_lr"], learning_rate_cooldown_iterations=cfg["cooldown_iterations"],
        learning_rate_cooldown_trigger_percent_threshold=cfg["cooldown_trigger_percent"])
    head = ace_network.Head(torch.from_numpy(prob["mean"]), nb, homog)
    P = head_oracle.HeadParams(flat0.clone(), nb, homog)
    sd = head.state_dict()
    for l, name in enumerate(head_oracle.head_layer_names(nb)):
        sd[name + ".weight"] = P.W[l].clone().view(512, 512, 1, 1)
        sd[name + ".bias"] = P.b[l].clone()
    sd["fc3.weight"] = P.W3.clone().view(4 if homog else 3, 512, 1, 1)
    sd["fc3.bias"] = P.b3.clone()
    head.load_state_dict(sd)
    head.train()

    tr = object.__new__(ace_trainer.TrainerACE)
    tr.options = opt
    tr.iteration = 0
    tr.epoch = 0
    tr.use_depth = bool(cfg["use_depth"])
    tr.iterations_output = 10 ** 9
    tr.ace_visualizer = None
    tr.training_start = time.time()
    tr.log_file = io.StringIO()
    tr.regressor = types.SimpleNamespace(get_scene_coordinates=head)
    tr.training_scheduler = ace_schedule.ScheduleACE(head, opt)
    tr.repro_loss = ace_loss.ReproLoss(total_iterations=cfg["iterations"], soft_clamp=cfg["soft_clamp"],
                                       soft_clamp_min=cfg["soft_clamp_min"], type=cfg["loss_type"], circle_schedule=True)

    class NoRefiner:  # pose_refinement 'none' (refine_poses.py:221-223)
        def get_current_poses(self, p, idx): return p.clone()
        def zero_grad(self, set_to_none=False): pass
        def step(self): pass
        def get_all_original_poses(self): return torch.zeros(1, 3, 4)
        def get_all_current_poses(self): return torch.zeros(1, 3, 4)
    tr.pose_refiner = NoRefiner()
    if cfg["pose_refinement"] in ("mlp", "naive"):
        n_img = prob["image_pose_inv"].shape[0]
        class FakeDS:
            poses = [torch.from_numpy(np.linalg.inv(prob["image_pose_inv"][i].astype(np.float64)).astype(np.float32)) for i in range(n_img)]
            def __len__(self): return n_img
        fake_ds = FakeDS()
        pr = refine_poses.PoseRefiner(fake_ds, torch.device("cpu"), opt)
        pr.create_pose_buffer()
        tr.pose_refiner = pr
    if cfg["pose_refinement"] == "mlp":
        from acezero_amd.head import init_pose_network, POSE_LAYERS
        flatp = init_pose_network(SEED + 3)
        sdp, o = {}, 0
        for lname, O, K in POSE_LAYERS:
            sdp[lname + ".weight"] = flatp[o:o + O * K].view(O, K, 1, 1).clone(); o += O * K
            sdp[lname + ".bias"] = flatp[o:o + O].clone(); o += O
        pr.pose_network.load_state_dict(sdp)
    if cfg["refine_calibration"]:
        ds = types.SimpleNamespace(get_focal_length=lambda i: float(prob["focal"]), __len__=lambda: 1)
        cr = object.__new__(refine_calibration.CalibrationRefiner)
        cr.focal_length_init = float(prob["focal"])
        cr.global_f = torch.zeros(1).detach().requires_grad_()
        cr.optimizer = torch.optim.AdamW([cr.global_f], lr=cfg["calib_lr"])
        tr.K_optimizer = cr
    else:
        tr.K_optimizer = None

    rec = {"loss": [], "inliers": [], "lr": [], "max_iterations": [], "focal_scale": [], "poses": [], "pose_params": []}
    sched = tr.training_scheduler
    orig_backward, orig_step = sched.backward, sched.step

    def backward(loss):
        rec["loss"].append(float(loss))
        orig_backward(loss)

    def step(batch_inliers):
        rec["inliers"].append(float(batch_inliers))
        rec["lr"].append(float(sched.optimizer.param_groups[0]["lr"]))
        orig_step(batch_inliers)
    sched.backward, sched.step = backward, step

    snaps = {}
    coords0 = None
    for it, idx in enumerate(batches):
        pp = synth.expand_per_patch(prob, idx)
        t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in pp.items()}
        if it == 0:
            with torch.no_grad():
                f = t["features"][None, None, ...].view(-1, 16, 32, 512).permute(0, 3, 1, 2)
                coords0 = head(f).permute(0, 2, 3, 1).flatten(0, 2).clone().numpy()
        n_before = len(rec["loss"])
        ace_trainer.TrainerACE.training_step(tr, t["features"], t["target_px"], t["aug_inv"], t["pose_inv"], t["K"], t["Kinv"],
                                             t["target_crds"] if cfg["use_depth"] else torch.zeros(B, 3), t["pose_idx"][:, None])
        ran = len(rec["loss"]) > n_before
        rec["max_iterations"].append(int(sched.max_iterations))
        rec["focal_scale"].append(float(1 + tr.K_optimizer.global_f) if tr.K_optimizer is not None else 1.0)
        if cfg["pose_refinement"] in ("mlp", "naive"):
            rec["poses"].append(tr.pose_refiner.get_all_current_poses().detach().numpy().copy())
            pp_ = tr.pose_refiner.pose_network.parameters() if cfg["pose_refinement"] == "mlp" else [tr.pose_refiner.pose_buffer]
            rec["pose_params"].append(torch.cat([p.detach().flatten() for p in pp_]).numpy().copy())
        if not ran:
            break
        tr.iteration += 1
        snap = torch.cat([p.detach().flatten() for p in head.parameters()]).numpy().copy()
        if it == 0:
            snaps[0] = snap
        snaps = {0: snaps[0], it: snap}
    return rec, snaps, coords0


def time_reference_step(batch=5120, budget_s=12.0, max_steps=40, threads=None):
    """Seconds per TrainerACE.training_step of the REFERENCE on this host's cores (CPU PyTorch fp32) at `batch` patches: bench.py's
    cpu_baseline, kind "reference", when /root/reference is present (the reference pins BLAS / OpenMP to one thread at import,
    ace_trainer.py:5-8; overridden here and stated with the number)."""
    global B
    if threads:
        torch.set_num_threads(threads)
    cfg = full_cfg(CONFIGS["head_tanh_1cyclepoly"])
    cfg.update(global_batch=batch, lr_min=0.0005, lr_max=0.003, warmup_iterations=1000, warmup_lr=0.0005, cooldown_iterations=5000,
               cooldown_trigger_percent=0.7, iterations=25000)
    prob = synth.make_training_problem(seed=3, n_images=20, views_per_image=2, patches_per_view=128)
    flat0 = head_oracle.init_params(1)
    rng = np.random.default_rng(0)
    idx = rng.integers(0, prob["features"].shape[0], batch)
    old_B, B = B, batch
    try:
        t_steps = []
        run_reference(cfg, prob, flat0, [idx])                     # warm-up (allocations, thread pools)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < budget_s and n < max_steps:
            t1 = time.perf_counter()
            run_reference(cfg, prob, flat0, [idx, idx][:1])
            t_steps.append(time.perf_counter() - t1)
            n += 1
    finally:
        B = old_B
    # run_reference rebuilds the head and the optimiser for every call; the step itself is the bulk (0.2-0.4 s of a call)
    return float(np.median(t_steps)), n


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    only = sys.argv[1:]   # optional: names of the configurations to (re)generate
    from tests import helpers   # the "trained regime" configurations and their problem live next to the tests that consume them
    global B
    for name, c in (list(CONFIGS.items()) + list(helpers.TRAINED_CONFIGS.items()) + [("head_focal_drift", helpers.FOCAL_DRIFT), ("head_trajectory", helpers.TRAJECTORY)] +
                    list(helpers.BIG_CONFIGS.items())):
        if only and name not in only:
            continue
        B = helpers.BIG_B if name in helpers.BIG_CONFIGS else 512
        if name in helpers.BIG_CONFIGS:
            prob, flat0, cfg = helpers.problem_for(name)
        elif name in helpers.TRAINED_CONFIGS or name in ("head_focal_drift", "head_trajectory"):
            prob, flat0, cfg = helpers.problem_for(name)
        else:
            cfg = full_cfg(c)
            prob = synth.make_training_problem(seed=SEED, n_images=6, views_per_image=2, patches_per_view=128)
            # the trainer stores features in half precision; keep them bf16-representable so both modes see the same inputs
            prob["features"] = torch.from_numpy(prob["features"]).to(torch.bfloat16).to(torch.float32).numpy()
            flat0 = head_oracle.init_params(SEED + 1)
        rng = np.random.default_rng(SEED + 2)
        n = prob["features"].shape[0]
        batches = helpers.big_batches(prob, cfg["steps"]) if name in helpers.BIG_CONFIGS else [rng.permutation(n)[:B] for _ in range(cfg["steps"])]
        rec, snaps, coords0 = run_reference(cfg, prob, flat0, batches)
        first = snaps[0]
        last_it = max(snaps)
        sel = np.arange(0, first.size, 997)
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            loss=np.array(rec["loss"], np.float64), inliers=np.array(rec["inliers"], np.float64), lr=np.array(rec["lr"], np.float64),
            max_iterations=np.array(rec["max_iterations"], np.int64), focal_scale=np.array(rec["focal_scale"], np.float64),
            coords0=coords0[:64].astype(np.float32), param_sel=sel, params_after_first=first[sel], params_after_last=snaps[last_it][sel],
            last_it=np.int64(last_it), steps_run=np.int64(len(rec["loss"])),
            poses=np.array(rec["poses"], np.float32),
            # (long trajectories: the sampled pose parameters of every 20th step only)
            pose_params_sel=np.array([p[::(97 if p.size > 1000 else 1)] for i, p in enumerate(rec["pose_params"]) if cfg["steps"] <= 50 or i % 20 == 19], np.float32))
        print(name, "steps run", len(rec["loss"]), "loss", rec["loss"][:3], "inl", rec["inliers"][:3], "lr", rec["lr"][:3],
              "max_it", rec["max_iterations"][-1], "focal", rec["focal_scale"][-1])


if __name__ == "__main__":
    main()

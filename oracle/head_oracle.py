"""CPU oracle of the ACE head training step -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(acezero_amd/) never does.  It restates, with plain torch-CPU tensor algebra and a hand-written backward pass,
what the reference computes in

    ace_network.py:120-149   Head.forward
    ace_trainer.py:499-679   TrainerACE.training_step (geometry, masks, loss / B)
    ace_loss.py:39-91        ReproLoss.compute
    ace_schedule.py:12-126   ScheduleACE (AdamW + LR schedules + cool-down trigger)
    refine_calibration.py:34-59

Three arithmetic modes:
  * "fp32": no rounding anywhere.  PINNED against the reference itself: tests/golden/head_*.npz were produced by
    running the reference's own TrainerACE.training_step (tests/golden/make_head_golden.py) and
    tests/test_head_oracle.py checks this oracle against them.
  * "fp16": the same rounding points in IEEE half precision (the reference's autocast format, ace_trainer.py:517-518), propagated
    gradients rounded after scaling by a power of two (the kernels' fp16 mode, acez_train_config.compute_dtype = ACEZ_DTYPE_FP16).
  * "bf16": operands of every 512-wide matmul (activations, weights, propagated gradients) are rounded to
    bfloat16 at the points where the HIP kernels store them; accumulation stays fp32.  This is what the GPU
    results are compared with (tolerance 1e-3 relative, BASELINE.json north_star).
"""
import math

import numpy as np
import torch

LOSS_TYPES = {"tanh": 0, "dyntanh": 1, "l1": 2, "l1+sqrt": 3, "l1+logl1": 4}


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def fp16_round(x):
    return x.to(torch.float16).to(torch.float32)


GRAD_SCALE_FP16 = 64.0     # a factor the HIP kernels put on the gradient chain in fp16 mode: a power of two chosen per step from the
                           # largest |d loss / d fc3 output| (head_kernels.hip sched_post_wave); scaling by a power of two commutes with fp16
                           # rounding wherever nothing under- or overflows, so ANY in-range power of two gives the same result


def head_layer_names(num_head_blocks):
    """state_dict prefixes of the 512-wide layers in Head.named_parameters() order (ace_network.py:85-107)."""
    names = ["res3_conv1", "res3_conv2", "res3_conv3"]
    for b in range(num_head_blocks):
        names += [f"{b}c0", f"{b}c1", f"{b}c2"]
    names += ["fc1", "fc2"]
    return names


def num_params(num_head_blocks, use_homogeneous=True):
    L = 3 + 3 * num_head_blocks + 2
    no = 4 if use_homogeneous else 3
    return L * (512 * 512 + 512) + no * 513


class HeadParams:
    """Flat fp32 parameter vector with (weight, bias) views per layer, same order as the C ABI."""

    def __init__(self, flat, num_head_blocks=1, use_homogeneous=True):
        self.nb = num_head_blocks
        self.L = 3 + 3 * num_head_blocks + 2
        self.no = 4 if use_homogeneous else 3
        assert flat.numel() == num_params(num_head_blocks, use_homogeneous)
        self.flat = flat
        self.W, self.b = [], []
        o = 0
        for _ in range(self.L):
            self.W.append(flat[o:o + 262144].view(512, 512))
            o += 262144
            self.b.append(flat[o:o + 512])
            o += 512
        self.W3 = flat[o:o + self.no * 512].view(self.no, 512)
        o += self.no * 512
        self.b3 = flat[o:o + self.no]


def init_params(seed, num_head_blocks=1, use_homogeneous=True, scale=1.0):
    """Deterministic (numpy PCG64) stand-in for nn.Conv2d's default init: U(-1/sqrt(512), 1/sqrt(512)).
    The generator lives with the other synthetic inputs (acezero_amd/synth.py) so that bench.py's timed legs need no oracle."""
    from acezero_amd import synth
    return torch.from_numpy(synth.init_head_params(seed, num_head_blocks, use_homogeneous, scale))


class HeadOracle:
    def __init__(self, flat_params, mean, num_head_blocks=1, use_homogeneous=True, mode="fp32",
                 homogeneous_min_scale=0.01, homogeneous_max_scale=4.0):
        assert mode in ("fp32", "bf16", "fp16")
        self.mode = mode
        self.r = bf16_round if mode == "bf16" else (fp16_round if mode == "fp16" else (lambda x: x))
        self.gs = GRAD_SCALE_FP16 if mode == "fp16" else 1.0
        self.absmax = 0.0     # largest scaled magnitude the last backward rounded (fp16: drives the next step's scale, as on the device)
        self.p = HeadParams(flat_params, num_head_blocks, use_homogeneous)
        self.nb = num_head_blocks
        self.homog = use_homogeneous
        self.mean = torch.as_tensor(mean, dtype=torch.float32).view(3)
        # buffers of ace_network.py:109-115 (float32 arithmetic like torch.tensor([..]))
        max_scale = torch.tensor([homogeneous_max_scale])
        min_scale = torch.tensor([homogeneous_min_scale])
        self.max_inv_scale = float((1.0 / max_scale).item())
        self.h_beta = float((math.log(2) / (1.0 - 1.0 / max_scale)).item())
        self.min_inv_scale = float((1.0 / min_scale).item())

    # ------------------------------------------------------------------ forward
    def forward(self, feats):
        """feats [B,512] fp32 (bf16-representable in bf16 mode). Returns fc3 output s [B,no] and the tape."""
        r, p = self.r, self.p
        tape = {"out": [None] * p.L, "R": [None] * (self.nb + 2)}
        R = r(feats)
        tape["R"][0] = R
        for b in range(self.nb + 1):
            l = 3 * b
            x1 = r(torch.relu(R @ r(p.W[l]).t() + p.b[l]))
            x2 = r(torch.relu(x1 @ r(p.W[l + 1]).t() + p.b[l + 1]))
            y3 = r(torch.relu(x2 @ r(p.W[l + 2]).t() + p.b[l + 2]))
            tape["out"][l], tape["out"][l + 1], tape["out"][l + 2] = x1, x2, y3
            R = r(y3 + R)  # ace_network.py:126,133
            tape["R"][b + 1] = R
        f1i = 3 * (self.nb + 1)
        f1 = r(torch.relu(R @ r(p.W[f1i]).t() + p.b[f1i]))
        f2 = r(torch.relu(f1 @ r(p.W[f1i + 1]).t() + p.b[f1i + 1]))
        tape["out"][f1i], tape["out"][f1i + 1] = f1, f2
        s = f2 @ r(p.W3).t() + p.b3
        return s, tape

    def dehomogenise(self, s):
        """ace_network.py:139-147. Returns X [B,3] and the pieces the backward needs."""
        if not self.homog:
            return s[:, :3] + self.mean, None
        s3 = s[:, 3]
        bx = self.h_beta * s3
        big = bx > 20.0   # F.softplus's threshold (ace_network.py:142). The unselected branch is evaluated on a harmless argument: under autograd
        sp = torch.where(big, s3, torch.log1p(torch.exp(torch.where(big, torch.zeros_like(bx), bx))) / self.h_beta)   # (pose path) inf * 0 would be NaN
        hraw = sp + self.max_inv_scale
        clamped = hraw > self.min_inv_scale
        h = torch.where(clamped, torch.full_like(hraw, self.min_inv_scale), hraw)
        X = s[:, :3] / h[:, None] + self.mean
        return X, (h, clamped, bx)

    def scene_coordinates(self, feats):
        s, _ = self.forward(feats)
        return self.dehomogenise(s)[0]

    # ------------------------------------------------------------------ loss + gradient wrt s
    def loss_and_ds(self, s, batch, cfg, iteration, focal_scale=1.0):
        """batch: dict with target_px [B,2], aug_inv [B,3,4], pose_inv [B,4,4], K [B,3,3], Kinv [B,3,3].
        Returns dict(loss_sum, inliers, ds [B,no], focal_grad, X)."""
        B = s.shape[0]
        invB = 1.0 / float(cfg["global_batch"])
        X, hh = self.dehomogenise(s)
        tu, tv = batch["target_px"][:, 0], batch["target_px"][:, 1]
        P = torch.bmm(batch["aug_inv"], batch["pose_inv"])  # [B,3,4]   ace_trainer.py:530
        Xh = torch.cat([X, torch.ones(B, 1)], dim=1)
        Xc = torch.bmm(P, Xh[:, :, None])[:, :, 0]  # :533
        K = batch["K"].clone()
        kscale = None
        if cfg.get("refine_calibration", False):
            f0 = float(cfg["focal_init"])
            kscale = K[:, 0, 0] / f0
            f = (np.float32(focal_scale) * np.float32(f0)) * kscale
            K[:, 0, 0] = f
            K[:, 0, 1] = 0.0
            K[:, 1, 0] = 0.0
            K[:, 1, 1] = f
        pp = torch.bmm(K, Xc[:, :, None])[:, :, 0]  # :536-540
        dmin, dmax = float(cfg["depth_min"]), float(cfg["depth_max"])
        zcl = pp[:, 2] < dmin
        pz = torch.where(zcl, torch.full_like(pp[:, 2], dmin), pp[:, 2])  # :545
        u, v = pp[:, 0] / pz, pp[:, 1] / pz
        du, dv = u - tu, v - tv
        e = du.abs() + dv.abs()  # :552
        invalid = (Xc[:, 2] < dmin) | (e > float(cfg["hard_clamp"])) | (Xc[:, 2] > dmax)  # :558-565
        use_depth = "target_crds" in batch and batch["target_crds"] is not None and cfg.get("use_depth", False)
        if use_depth:   # ace_trainer.py:567-574
            tcd = batch["target_crds"] - X
            tdist = torch.linalg.norm(tcd, dim=1)
            tavail = batch["target_crds"].abs().sum(dim=1) > 0.00001
            invalid = invalid | ((tdist > 0.1) & tavail)
        valid = ~invalid
        # loss weight (ace_loss.py:55-69)
        lt = cfg["loss_type"]
        w = float(cfg["soft_clamp"])
        if lt == "dyntanh":
            sw = iteration / float(cfg["iterations"])
            if cfg.get("circle_schedule", True):
                sw = 1 - np.sqrt(1 - sw ** 2)
            w = float(np.float32((1 - sw) * cfg["soft_clamp"] + cfg["soft_clamp_min"]))
        if lt in ("tanh", "dyntanh"):
            th = torch.tanh(e / w)
            lv = w * th
            ge = 1 - th * th
        elif lt == "l1":
            big = e > w
            lv = torch.where(big, torch.zeros_like(e), e)
            ge = torch.where(big, torch.zeros_like(e), torch.ones_like(e))
        elif lt == "l1+sqrt":
            big = e > w
            lv = torch.where(big, torch.sqrt(w * e), e)
            ge = torch.where(big, 0.5 * w / torch.sqrt(w * e), torch.ones_like(e))
        else:
            big = e > w
            lv = torch.where(big, torch.log(1 + w * e), e)
            ge = torch.where(big, w / (1 + w * e), torch.ones_like(e))
        inl = (valid & (e < float(cfg["inlier_px_threshold"]))).float()
        gu, gv = ge * torch.sign(du), ge * torch.sign(dv)
        dp = torch.stack([gu / pz, gv / pz, torch.where(zcl, torch.zeros_like(pz), -(gu * pp[:, 0] + gv * pp[:, 1]) / (pz * pz))], dim=1)
        dXc_valid = torch.bmm(K.transpose(1, 2), dp[:, :, None])[:, :, 0]
        # invalid branch: proxy target at constant depth (ace_trainer.py:592-600)
        px_h = torch.stack([tu, tv, torch.ones_like(tu)], dim=1)
        tgt = float(cfg["depth_target"]) * torch.bmm(batch["Kinv"], px_h[:, :, None])[:, :, 0]
        d = tgt - Xc
        li = d.abs().sum(dim=1)
        dXc_inv = -torch.sign(d)
        dXs = torch.zeros_like(X)
        if use_depth:   # ace_trainer.py:601-609: L2 distance to the GT coordinate where available, nothing otherwise
            li = torch.where(tavail, tdist, torch.zeros_like(tdist))
            dXc_inv = torch.zeros_like(dXc_inv)
            inv = torch.where(tdist > 0, 1.0 / tdist, torch.zeros_like(tdist))
            dXs = torch.where((invalid & tavail)[:, None], -tcd * inv[:, None], dXs)
        loss_rows = torch.where(valid, lv, li)
        dXc = torch.where(valid[:, None], dXc_valid, dXc_inv)
        dX = (torch.bmm(P[:, :, :3].transpose(1, 2), dXc[:, :, None])[:, :, 0] + dXs) * invB
        focal_grad = 0.0
        if kscale is not None:
            fg = torch.where(valid, (dp[:, 0] * Xc[:, 0] + dp[:, 1] * Xc[:, 1]) * float(cfg["focal_init"]) * kscale, torch.zeros_like(e))
            focal_grad = float((fg * invB).sum())
        if self.homog:
            h, clamped, bx = hh
            ds = torch.zeros(B, 4)
            ds[:, :3] = dX / h[:, None]
            dh = -(dX * s[:, :3]).sum(dim=1) / (h * h)
            dh = torch.where(clamped, torch.zeros_like(dh), dh)
            z = torch.exp(bx)
            ds[:, 3] = torch.where(bx > 20.0, dh, dh * z / (z + 1))
        else:
            ds = dX.clone()
        return {"loss_sum": float(loss_rows.sum()), "inliers": float(inl.sum()), "ds": ds, "focal_grad": focal_grad,
                "X": X, "e": e, "valid": valid}

    # ------------------------------------------------------------------ backward through the MLP
    def rg(self, g):
        """rounding of a propagated gradient (stored scaled by self.gs in fp16 mode)"""
        if self.mode == "fp16":
            self.absmax = max(self.absmax, float((g * self.gs).abs().max()))
        return self.r(g * self.gs) / self.gs

    def next_grad_scale(self):
        """head_kernels.hip sched_post_wave: the power of two that would have put the step's largest propagated gradient near 4096"""
        if self.mode == "fp16" and self.absmax > 0:
            e = math.floor(math.log2(4096.0 / (self.absmax / self.gs)))
            self.gs = 2.0 ** min(16, max(-8, e))
        self.absmax = 0.0

    def backward(self, tape, ds):
        """Returns the flat gradient (same layout as the parameters)."""
        r, rg, p = self.r, self.rg, self.p
        g = torch.zeros_like(p.flat)
        G = HeadParams(g, self.nb, self.homog)
        f1i = 3 * (self.nb + 1)
        f2 = tape["out"][f1i + 1]
        G.W3.copy_(ds.t() @ f2)
        G.b3.copy_(ds.sum(dim=0))
        dZ = [None] * p.L
        dZ[f1i + 1] = rg((ds @ r(p.W3)) * (f2 > 0))
        dZ[f1i] = rg(dZ[f1i + 1] @ r(p.W[f1i + 1])) * (tape["out"][f1i] > 0)
        t = rg(dZ[f1i] @ r(p.W[f1i]))
        dR = t
        dZ[3 * self.nb + 2] = t * (tape["out"][3 * self.nb + 2] > 0)
        for b in range(self.nb, -1, -1):
            l = 3 * b
            dZ[l + 1] = rg(dZ[l + 2] @ r(p.W[l + 2])) * (tape["out"][l + 1] > 0)
            dZ[l] = rg(dZ[l + 1] @ r(p.W[l + 1])) * (tape["out"][l] > 0)
            if b > 0:
                t = rg(dZ[l] @ r(p.W[l]) + dR)
                dR = t
                dZ[l - 1] = t * (tape["out"][l - 1] > 0)
        ins = [None] * p.L
        for b in range(self.nb + 1):
            ins[3 * b] = tape["R"][b]
            ins[3 * b + 1] = tape["out"][3 * b]
            ins[3 * b + 2] = tape["out"][3 * b + 1]
        ins[f1i] = tape["R"][self.nb + 1]
        ins[f1i + 1] = tape["out"][f1i]
        for l in range(p.L):
            G.W[l].copy_(dZ[l].t() @ ins[l])
            G.b[l].copy_(dZ[l].sum(dim=0))
        return g


POSE_LAYERS = [("head_skip", 128, 12), ("conv1", 128, 12), ("conv2", 128, 128), ("conv3", 128, 128), ("fc1", 128, 128),
               ("fc2", 128, 128), ("fc3", 12, 128)]   # PoseNetwork(0, 128).named_parameters() order (refine_poses.py:21-51)


def special_gramschmidt(M):
    """roma.special_gramschmidt (roma 1.4.1, not vendored in the reference: restated from its documentation --
    Gram-Schmidt on the first two columns, third column = cross product)."""
    x, y = M[..., 0], M[..., 1]
    x = x / torch.norm(x, dim=-1, keepdim=True)
    y = y - torch.sum(x * y, dim=-1, keepdim=True) * x
    y = y / torch.norm(y, dim=-1, keepdim=True)
    z = torch.cross(x, y, dim=-1)
    return torch.stack((x, y, z), dim=-1)


def special_procrustes(M):
    """roma.special_procrustes (roma 1.4.1, not vendored: restated from its documentation -- the rotation closest to M in the
    Frobenius norm, R = U diag(1, 1, det(U V^T)) V^T for the SVD M = U S V^T)."""
    U, _, Vh = torch.linalg.svd(M)
    d = torch.det(U @ Vh)
    D = torch.diag_embed(torch.stack([torch.ones_like(d), torch.ones_like(d), d], dim=-1))
    return U @ D @ Vh


def orthonormalise(M, ortho):
    return special_procrustes(M) if ortho == "procrustes" else special_gramschmidt(M)


class PoseMLPOracle:
    """refine_poses.py `mlp` strategy evaluated once per image, torch autograd for the backward pass."""

    def __init__(self, flat, update_weight=0.1, ortho="gram-schmidt"):
        self.flat = flat.clone().requires_grad_(True)
        self.w = update_weight
        self.ortho = ortho

    def _views(self):
        out, o = {}, 0
        for name, O, K in POSE_LAYERS:
            out[name + ".weight"] = self.flat[o:o + O * K].view(O, K); o += O * K
            out[name + ".bias"] = self.flat[o:o + O]; o += O
        return out

    def forward(self, poses44):
        """poses44 [I,4,4] world->cam -> refined [I,4,4] (refine_poses.py:53-72,152-176,236-244)."""
        v = self._views()
        x0 = poses44[:, :3].reshape(-1, 12)
        lin = lambda x, n: x @ v[n + ".weight"].t() + v[n + ".bias"]
        x = torch.relu(lin(x0, "conv1")); x = torch.relu(lin(x, "conv2")); x = torch.relu(lin(x, "conv3"))
        res = lin(x0, "head_skip") + x
        d = lin(torch.relu(lin(torch.relu(lin(res, "fc1")), "fc2")), "fc3")
        upd = (x0 + self.w * d).view(-1, 3, 4)
        R = orthonormalise(upd[:, :3, :3], self.ortho)
        out = poses44.clone()
        out = torch.cat([torch.cat([R, upd[:, :3, 3:4]], dim=2), poses44[:, 3:4, :]], dim=1)
        return out


class PoseNaiveOracle:
    """refine_poses.py `naive` strategy (:224-234): the [I,3,4] world->cam poses are the parameters."""

    def __init__(self, flat, ortho="gram-schmidt"):
        self.flat = flat.clone().requires_grad_(True)
        self.ortho = ortho

    def forward(self, poses44):
        cur = self.flat.view(-1, 3, 4)
        R = orthonormalise(cur[:, :3, :3], self.ortho)
        return torch.cat([torch.cat([R, cur[:, :3, 3:4]], dim=2), poses44[:, 3:4, :]], dim=1)


def loss_autograd(head, s, batch, cfg, iteration, poses44_rows, focal_g=None):
    """ace_trainer.py:521-613 with torch ops and autograd (used where gradients wrt the poses are needed).
    s [B,no] requires grad; poses44_rows [B,4,4] may require grad; focal_g: the CalibrationRefiner's scalar (a tensor that requires grad,
    refine_calibration.py:20-53) when cfg["refine_calibration"]. Returns (loss_sum tensor, inliers)."""
    B = s.shape[0]
    X, _ = head.dehomogenise(s)
    tu_tv = batch["target_px"]
    P = torch.bmm(batch["aug_inv"], poses44_rows)
    Xh = torch.cat([X, torch.ones(B, 1)], dim=1)
    Xc = torch.bmm(P, Xh[:, :, None])[:, :, 0]
    K = batch["K"].clone()
    if cfg.get("refine_calibration", False):
        # refine_calibration.py:34-53: f = (1 + g) f0, times the per-patch augmentation scale K00 / f0 (detached), written over K[:2,:2]
        f0 = float(cfg["focal_init"])
        f = ((1.0 + focal_g) * np.float32(f0)) * (batch["K"][:, 0, 0] / f0)
        zero = torch.zeros_like(f)
        K = torch.stack([torch.stack([f, zero, K[:, 0, 2]], dim=1), torch.stack([zero, f, K[:, 1, 2]], dim=1), K[:, 2]], dim=1)
    pp = torch.bmm(K, Xc[:, :, None])[:, :, 0]
    pz = pp[:, 2].clamp(min=float(cfg["depth_min"]))
    uv = pp[:, :2] / pz[:, None]
    e = (uv - tu_tv).abs().sum(dim=1)
    invalid = (Xc[:, 2] < cfg["depth_min"]) | (e > cfg["hard_clamp"]) | (Xc[:, 2] > cfg["depth_max"])
    valid = ~invalid
    w = float(cfg["soft_clamp"])
    assert cfg["loss_type"] == "tanh"
    lv = w * torch.tanh(e / w)
    px_h = torch.cat([tu_tv, torch.ones(B, 1)], dim=1)
    tgt = float(cfg["depth_target"]) * torch.bmm(batch["Kinv"], px_h[:, :, None])[:, :, 0]
    li = (tgt - Xc).abs().sum(dim=1)
    loss_sum = torch.where(valid, lv, li).sum()
    inl = float((valid & (e < cfg["inlier_px_threshold"])).sum())
    return loss_sum, inl


class ScheduleOracle:
    """ace_schedule.py restated with the chainable LR recurrences of torch.optim.lr_scheduler (LinearLR,
    OneCycleLR) in Python floats, plus single-tensor AdamW (torch.optim.AdamW defaults)."""

    def __init__(self, cfg, n_params):
        self.c = cfg
        self.schedule = cfg["schedule"]
        self.max_iterations = cfg["iterations"]
        self.in_cooldown = False
        self.buf = []
        self.warm_epoch = 0
        self.cool_epoch = 0
        self.steps = 0
        self.m = torch.zeros(n_params)
        self.v = torch.zeros(n_params)
        if self.schedule == "constant":
            self.lr = cfg["lr_min"]
        elif self.schedule == "1cyclepoly":
            self.lr = cfg["lr_max"] * (cfg["warmup_lr"] / cfg["lr_max"])
        else:
            self.lr = self._onecycle(0)
        self.beta1, self.beta2, self.eps, self.wd = 0.9, 0.999, 1e-8, 1e-2
        # calibration refiner state
        self.calib_g, self.calib_m, self.calib_v, self.calib_steps = 0.0, 0.0, 0.0, 0

    def _onecycle(self, step_num):
        c = self.c
        initial_lr = c["lr_max"] / 25.0
        min_lr = initial_lr / 1e4
        end1 = float(0.3 * c["iterations"]) - 1
        end2 = c["iterations"] - 1
        if step_num <= end1:
            start, end, pct = initial_lr, c["lr_max"], (step_num / end1 if end1 > 0 else 1.0)
        else:
            start, end, pct = c["lr_max"], min_lr, (step_num - end1) / (end2 - end1)
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1)

    def check_and_set_cooldown(self, iteration):
        c = self.c
        if self.schedule != "1cyclepoly" or self.in_cooldown or iteration < c["warmup_iterations"]:
            return
        by_duration = iteration >= (self.max_iterations - c["cooldown_iterations"])
        dynamic = len(self.buf) > 0 and min(self.buf) > c["cooldown_trigger_percent"]
        if by_duration or dynamic:
            self.in_cooldown = True
            self.cool_epoch = 0
            self.max_iterations = iteration + c["cooldown_iterations"]

    def adamw(self, p, g):
        self.steps += 1
        lr = self.lr
        p.mul_(1 - lr * self.wd)
        self.m.lerp_(g, 1 - self.beta1)
        self.v.mul_(self.beta2).addcmul_(g, g, value=1 - self.beta2)
        bc1 = 1 - self.beta1 ** self.steps
        bc2 = 1 - self.beta2 ** self.steps
        denom = (self.v.sqrt() / math.sqrt(bc2)).add_(self.eps)
        p.addcdiv_(self.m, denom, value=-(lr / bc1))

    def calib_step(self, g):
        lr = self.c["calib_lr"]
        self.calib_steps += 1
        f32 = lambda x: float(np.float32(x))
        p = self.calib_g * (1 - lr * self.wd)
        self.calib_m = self.calib_m + (g - self.calib_m) * (1 - self.beta1)
        self.calib_v = self.calib_v * self.beta2 + (1 - self.beta2) * g * g
        bc1 = 1 - self.beta1 ** self.calib_steps
        bc2 = 1 - self.beta2 ** self.calib_steps
        denom = math.sqrt(self.calib_v) / math.sqrt(bc2) + self.eps
        p = p - (lr / bc1) * (self.calib_m / denom)
        self.calib_g, self.calib_m, self.calib_v = f32(p), f32(self.calib_m), f32(self.calib_v)

    def sched_step(self, batch_inliers):
        c = self.c
        if self.schedule == "1cyclepoly":
            if self.in_cooldown:
                self.cool_epoch += 1
                e, sf, ef, T = self.cool_epoch, 1.0, c["lr_min"] / c["lr_max"], c["cooldown_iterations"]
                if e <= T:
                    self.lr = self.lr * (1.0 + (ef - sf) / (T * sf + (e - 1) * (ef - sf)))
            else:
                self.warm_epoch += 1
                e, sf, ef, T = self.warm_epoch, c["warmup_lr"] / c["lr_max"], 1.0, c["warmup_iterations"]
                if e <= T:
                    self.lr = self.lr * (1.0 + (ef - sf) / (T * sf + (e - 1) * (ef - sf)))
            self.buf.append(batch_inliers)
            if len(self.buf) > 100:
                self.buf = self.buf[1:]
        elif self.schedule == "circle":
            self.warm_epoch += 1
            self.lr = self._onecycle(self.warm_epoch)


class TrainerOracle:
    """training_step of ace_trainer.py:499-679 on explicit per-patch inputs."""

    def __init__(self, flat_params, mean, cfg, mode="fp32", pose_flat=None, image_pose_inv=None):
        self.cfg = cfg
        self.head = HeadOracle(flat_params, mean, cfg.get("num_head_blocks", 1), cfg.get("use_homogeneous", True), mode)
        self.sched = ScheduleOracle(cfg, flat_params.numel())
        self.iteration = 0
        self.log = []
        self.pose = None
        if cfg.get("pose_refinement", "none") in ("mlp", "naive"):
            ortho = cfg.get("refinement_ortho", "gram-schmidt")
            self.pose = (PoseMLPOracle(pose_flat, cfg.get("pose_refinement_weight", 0.1), ortho) if cfg["pose_refinement"] == "mlp"
                         else PoseNaiveOracle(pose_flat, ortho))
            self.image_pose_inv = torch.as_tensor(image_pose_inv, dtype=torch.float32)
            self.pose_m = torch.zeros_like(pose_flat)
            self.pose_v = torch.zeros_like(pose_flat)
            self.pose_steps = 0

    def current_poses(self):
        with torch.no_grad():
            return self.pose.forward(self.image_pose_inv)[:, :3].clone()

    def _pose_adamw(self, g, lr=0.001):
        self.pose_steps += 1
        b1, b2, eps, wd = 0.9, 0.999, 1e-8, 1e-2
        with torch.no_grad():
            p = self.pose.flat
            p.mul_(1 - lr * wd)
            self.pose_m.lerp_(g, 1 - b1)
            self.pose_v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (self.pose_v.sqrt() / math.sqrt(1 - b2 ** self.pose_steps)).add_(eps)
            p.addcdiv_(self.pose_m, denom, value=-(lr / (1 - b1 ** self.pose_steps)))

    def step(self, feats, batch):
        cfg, sch = self.cfg, self.sched
        sch.check_and_set_cooldown(self.iteration)
        if self.iteration >= sch.max_iterations:
            return None
        s, tape = self.head.forward(feats)
        pose_grad = None
        if self.pose is None:
            out = self.head.loss_and_ds(s, batch, cfg, self.iteration, focal_scale=1.0 + sch.calib_g)
        else:
            # refined poses (refine_poses.py:236-244), gradients wrt the fc3 outputs and the pose network by autograd
            self.pose.flat.grad = None
            poses_all = self.pose.forward(self.image_pose_inv)
            rows = poses_all[batch["pose_idx"].long().view(-1)]
            sl = s.detach().clone().requires_grad_(True)
            fg = torch.tensor(float(sch.calib_g), dtype=torch.float32, requires_grad=True) if cfg.get("refine_calibration", False) else None
            loss_sum, inl = loss_autograd(self.head, sl, batch, cfg, self.iteration, rows, focal_g=fg)
            (loss_sum / cfg["global_batch"]).backward()
            pose_grad = self.pose.flat.grad.detach().clone()
            with torch.no_grad():
                X = self.head.dehomogenise(sl)[0]
            out = {"loss_sum": float(loss_sum.detach()), "inliers": inl, "ds": sl.grad.detach(), "focal_grad": float(fg.grad) if fg is not None else 0.0,
                   "X": X.detach()}
        grad = self.head.backward(tape, out["ds"])
        self.head.next_grad_scale()   # (fp16 mode: the device adapts its gradient scale after every step; no-op otherwise)
        loss = out["loss_sum"] / cfg["global_batch"]
        inl = out["inliers"] / cfg["global_batch"]
        lr_used = sch.lr
        sch.adamw(self.head.p.flat, grad)
        if self.pose is not None and self.iteration > cfg.get("pose_refinement_wait", 0):   # ace_trainer.py:634
            self._pose_adamw(pose_grad, cfg.get("pose_refinement_lr", 0.001))
        if cfg.get("refine_calibration", False):
            sch.calib_step(out["focal_grad"])
        sch.sched_step(inl)
        rec = {"iteration": self.iteration, "loss": loss, "inliers": inl, "lr": lr_used, "grad": grad, "X": out["X"], "pose_grad": pose_grad}
        self.log.append({k: rec[k] for k in ("iteration", "loss", "inliers", "lr")})
        self.iteration += 1
        return rec

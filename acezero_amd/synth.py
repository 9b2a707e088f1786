"""Synthetic inputs of the hot path: no dataset, encoder weights or network are available,
so bench.py, smoke() and the parity tests run on seeded synthetic scenes.

  make_training_problem  a training buffer in the de-duplicated layout of include/acez.h (acez_train_buffer)
  make_registration_frames  60x80 scene-coordinate maps with noise and outliers + ground-truth poses

Everything is numpy (PCG64 streams are stable across numpy versions), so fixtures can be regenerated anywhere.
"""
import math

import numpy as np


def _rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def random_cameras(rng, n, room=(6.0, 4.0, 3.0)):
    """cam->world 4x4 poses: cameras inside a box room, looking roughly along +z of their own frame."""
    poses = np.zeros((n, 4, 4))
    for i in range(n):
        c = (rng.uniform(0.3, 0.7, size=3)) * np.array(room)
        R = _rot_xyz(rng.uniform(-0.3, 0.3), rng.uniform(-math.pi, math.pi), rng.uniform(-0.2, 0.2))
        poses[i, :3, :3] = R
        poses[i, :3, 3] = c
        poses[i, 3, 3] = 1.0
    return poses


def make_training_problem(seed=2089, n_images=10, views_per_image=2, patches_per_view=256, focal=525.0,
                          height=480, width=640, feature_noise=0.5):
    """Returns a dict of numpy arrays (float32 / int32) describing a training buffer.

    Geometry follows ace_trainer.py:372-436: a view is one (image, augmentation pass); a patch is one sampled
    encoder output location of that view with target pixel 8*(idx+0.5).
    """
    rng = np.random.default_rng(seed)
    cams = random_cameras(rng, n_images)
    n_views = n_images * views_per_image
    n = n_views * patches_per_view
    view_image = np.repeat(np.arange(n_images, dtype=np.int32), views_per_image)
    view_aug_inv = np.zeros((n_views, 3, 4), np.float32)
    view_K = np.zeros((n_views, 3, 3), np.float32)
    view_Kinv = np.zeros((n_views, 3, 3), np.float32)
    image_pose_inv = np.linalg.inv(cams).astype(np.float32)
    gh, gw = height // 8, width // 8
    target_px = np.zeros((n, 2), np.float32)
    view_idx = np.repeat(np.arange(n_views, dtype=np.int32), patches_per_view)
    gt = np.zeros((n, 3), np.float32)
    for v in range(n_views):
        ang = math.radians(rng.uniform(-15.0, 15.0))          # train_ace.py:178 aug_rotation 15
        scale = rng.uniform(2.0 / 3.0, 1.5)                   # train_ace.py:181-182
        Rz = _rot_xyz(0, 0, ang)
        view_aug_inv[v, :, :3] = Rz.T
        K = np.array([[focal * scale, 0, width / 2.0], [0, focal * scale, height / 2.0], [0, 0, 1.0]])
        view_K[v] = K
        view_Kinv[v] = np.linalg.inv(K)
        ys = rng.integers(0, gh, size=patches_per_view)
        xs = rng.integers(0, gw, size=patches_per_view)
        px = np.stack([8.0 * (xs + 0.5), 8.0 * (ys + 0.5)], axis=1)
        sl = slice(v * patches_per_view, (v + 1) * patches_per_view)
        target_px[sl] = px
        depth = rng.uniform(1.0, 5.0, size=patches_per_view)
        xc_aug = (np.linalg.inv(K) @ np.concatenate([px, np.ones((patches_per_view, 1))], axis=1).T) * depth  # 3 x P
        xc = Rz @ xc_aug
        cam = cams[view_image[v]]
        gt[sl] = (cam[:3, :3] @ xc + cam[:3, 3:4]).T
    mean = gt.mean(axis=0).astype(np.float32)
    proj = rng.normal(0, 1.0, size=(3, 512))
    feats = np.tanh((gt - mean) @ proj * 0.5) + feature_noise * rng.normal(0, 1.0, size=(n, 512))
    return {
        "features": feats.astype(np.float32), "target_px": target_px, "view_idx": view_idx,
        "view_aug_inv": view_aug_inv, "view_K": view_K, "view_Kinv": view_Kinv, "view_image": view_image,
        "image_pose_inv": image_pose_inv, "mean": mean, "gt_coords": gt, "focal": np.float32(focal),
        # ground-truth scene coordinates as the depth-based targets of ace_trainer.py:338 (a quarter of them "missing" = zeros)
        "target_crds": np.where((np.arange(n) % 4 == 3)[:, None], 0.0, gt).astype(np.float32),
    }


def expand_per_patch(prob, idx):
    """Per-patch replicated tensors, the layout ace_trainer.py:330-340 stores and training_step receives."""
    v = prob["view_idx"][idx]
    img = prob["view_image"][v]
    return {
        "features": prob["features"][idx], "target_px": prob["target_px"][idx], "aug_inv": prob["view_aug_inv"][v],
        "pose_inv": prob["image_pose_inv"][img], "K": prob["view_K"][v], "Kinv": prob["view_Kinv"][v],
        "pose_idx": img.astype(np.int16), "target_crds": prob["target_crds"][idx],
    }


ENCODER_LAYERS = [  # (name, c_in, c_out, kernel) in Encoder.__init__ order, ace_network.py:26-40
    ("conv1", 1, 32, 3), ("conv2", 32, 64, 3), ("conv3", 64, 128, 3), ("conv4", 128, 256, 3), ("res1_conv1", 256, 256, 3),
    ("res1_conv2", 256, 256, 1), ("res1_conv3", 256, 256, 3), ("res2_conv1", 256, 512, 3), ("res2_conv2", 512, 512, 1),
    ("res2_conv3", 512, 512, 3), ("res2_skip", 256, 512, 1)]


def init_encoder_weights(seed=4099, out_channels=512):
    """Seeded numpy stand-in for the pretrained encoder blob (not available offline): the reference's state_dict keys with
    nn.Conv2d's default init bounds U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias. Returns {key: float32 array}."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, ci, co, k in ENCODER_LAYERS:
        if name in ("res2_conv3", "res2_skip"):
            co = out_channels
        bound = 1.0 / np.sqrt(ci * k * k)
        sd[name + ".weight"] = rng.uniform(-bound, bound, size=(co, ci, k, k)).astype(np.float32)
        sd[name + ".bias"] = rng.uniform(-bound, bound, size=(co,)).astype(np.float32)
    return sd


def head_num_params(num_head_blocks=1, use_homogeneous=True):
    return (3 + 3 * num_head_blocks + 2) * (512 * 512 + 512) + (4 if use_homogeneous else 3) * 513


def init_head_params(seed, num_head_blocks=1, use_homogeneous=True, scale=1.0):
    """Deterministic (numpy PCG64) stand-in for nn.Conv2d's default init of the head: flat float32 vector in the order of
    Head.named_parameters() (the layout of acez_param_buffers.d_params)."""
    rng = np.random.default_rng(seed)
    bound = scale / math.sqrt(512.0)
    return rng.uniform(-bound, bound, size=head_num_params(num_head_blocks, use_homogeneous)).astype(np.float32)


def head_state_dict(flat, num_head_blocks=1, use_homogeneous=True, mean=(0.0, 0.0, 0.0)):
    """Head.state_dict()-shaped dict (numpy views) of a flat parameter vector (ace_network.py:85-118 key names)."""
    names = ["res3_conv1", "res3_conv2", "res3_conv3"]
    for b in range(num_head_blocks):
        names += [f"{b}c0", f"{b}c1", f"{b}c2"]
    names += ["fc1", "fc2"]
    flat = np.asarray(flat, np.float32)
    sd, o = {}, 0
    for name in names:
        sd[name + ".weight"] = flat[o:o + 262144].reshape(512, 512, 1, 1); o += 262144
        sd[name + ".bias"] = flat[o:o + 512]; o += 512
    no = 4 if use_homogeneous else 3
    sd["fc3.weight"] = flat[o:o + no * 512].reshape(no, 512, 1, 1); o += no * 512
    sd["fc3.bias"] = flat[o:o + no]
    sd["mean"] = np.asarray(mean, np.float32).reshape(1, 3, 1, 1)
    return sd


def make_gray_images(seed=77, n=2, h=480, w=640):
    """Normalised grayscale frames [n,1,h,w] float32 as the dataset hands them to the network: smooth random texture,
    (x/255 - 0.4) / 0.25 (dataset.py:150-153)."""
    rng = np.random.default_rng(seed)
    ys = np.arange(h, dtype=np.float64)[:, None]
    xs = np.arange(w, dtype=np.float64)[None, :]
    out = np.zeros((n, 1, h, w), np.float32)
    for i in range(n):
        img = np.zeros((h, w))
        for _ in range(12):
            fx, fy = rng.uniform(0.005, 0.25, size=2)
            ph = rng.uniform(0, 2 * math.pi)
            img += rng.uniform(0.3, 1.0) * np.sin(2 * math.pi * (fx * xs + fy * ys) + ph)
        img = (img - img.min()) / (img.max() - img.min() + 1e-9)          # [0,1]
        img = np.round(img * 255.0) / 255.0 + 0.0                        # 8-bit grey levels
        out[i, 0] = ((img - 0.4) / 0.25).astype(np.float32)
    return out


def make_registration_frames(seed=1305, n_frames=8, h=60, w=80, focal=525.0, subsampling=8, noise_sigma=0.02,
                             outlier_ratio=0.3):
    """Scene-coordinate maps [n,3,h,w] float32 for cameras in a box room + ground-truth cam->world poses.

    Each pixel (x*8+4, y*8+4) sees the point where its viewing ray hits the room box; a fraction of the pixels
    is replaced by uniform outliers inside the room, the rest gets Gaussian noise (SURVEY.md section 8d "R inputs").
    """
    rng = np.random.default_rng(seed)
    room = np.array([6.0, 4.0, 3.0])
    cams = random_cameras(rng, n_frames, tuple(room))
    ppx, ppy = w * subsampling / 2.0, h * subsampling / 2.0
    xs = (np.arange(w) * subsampling + subsampling // 2).astype(np.float64)
    ys = (np.arange(h) * subsampling + subsampling // 2).astype(np.float64)
    gx, gy = np.meshgrid(xs, ys)  # [h,w]
    rays = np.stack([(gx - ppx) / focal, (gy - ppy) / focal, np.ones_like(gx)], axis=0).reshape(3, -1)
    sc = np.zeros((n_frames, 3, h, w), np.float32)
    for i in range(n_frames):
        R, c = cams[i, :3, :3], cams[i, :3, 3]
        d = R @ rays  # world directions
        # distance to the box walls along each ray
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (0.0 - c[:, None]) / d
            t2 = (room[:, None] - c[:, None]) / d
        t = np.where(d > 0, t2, t1)
        t = np.where(np.abs(d) < 1e-12, np.inf, t)
        tmin = t.min(axis=0)
        pts = c[:, None] + d * tmin
        pts = pts + rng.normal(0, noise_sigma, size=pts.shape)
        out = rng.uniform(size=pts.shape[1]) < outlier_ratio
        pts[:, out] = rng.uniform(0, 1, size=(3, int(out.sum()))) * room[:, None]
        sc[i] = pts.reshape(3, h, w).astype(np.float32)
    return {"scene_coords": sc, "poses": cams, "focal": float(focal), "ppx": float(ppx), "ppy": float(ppy)}

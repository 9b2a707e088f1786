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
                          height=480, width=640, feature_noise=0.5, feature_gain=0.5):
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
    feats = np.tanh((gt - mean) @ proj * feature_gain) + feature_noise * rng.normal(0, 1.0, size=(n, 512))
    return {
        "features": feats.astype(np.float32), "target_px": target_px, "view_idx": view_idx,
        "view_aug_inv": view_aug_inv, "view_K": view_K, "view_Kinv": view_Kinv, "view_image": view_image,
        "image_pose_inv": image_pose_inv, "mean": mean, "gt_coords": gt, "focal": np.float32(focal), "feature_proj": proj,
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


def init_encoder_weights_bandpass(seed=4099, out_channels=512, out_scale=2.0):
    """A second stand-in for the pretrained encoder, used where the features have to be USABLE (end-to-end mapping /
    relocalisation tests): He-normal filters with their mean removed and zero biases. Default-initialised filters
    (init_encoder_weights) leave a dominant common-mode component after eleven ReLU layers (mean pairwise cosine of the
    pixel features 0.89, effective rank 18); zero-mean filters keep the features decorrelated (cosine 0.1-0.2, rank ~200).
    The filters of the three stride-2 layers are additionally smoothed with a 3x3 binomial kernel (then re-centred, energy
    kept): un-smoothed, the stride-8 sampling makes the features of the same scene point differ completely after a 2-pixel
    image shift (cosine 0.38), smoothed they stay similar (0.80) while unrelated points stay apart (0.23).
    The last two layers are scaled so that the feature magnitude is O(1)."""
    rng = np.random.default_rng(seed)
    k1 = np.array([1.0, 2.0, 1.0]) / 4.0
    B = np.outer(k1, k1)
    sd = {}
    for name, ci, co, k in ENCODER_LAYERS:
        if name in ("res2_conv3", "res2_skip"):
            co = out_channels
        w = rng.normal(0.0, np.sqrt(2.0 / (ci * k * k)), size=(co, ci, k, k))
        w -= w.mean(axis=(1, 2, 3), keepdims=True)
        if name in ("conv2", "conv3", "conv4"):
            e0 = np.sqrt((w ** 2).sum(axis=(1, 2, 3), keepdims=True))
            wp = np.pad(w, ((0, 0), (0, 0), (1, 1), (1, 1)))
            w = sum(B[a, b] * wp[:, :, a:a + 3, b:b + 3] for a in range(3) for b in range(3))
            w -= w.mean(axis=(1, 2, 3), keepdims=True)
            w *= e0 / np.sqrt((w ** 2).sum(axis=(1, 2, 3), keepdims=True))
        if name in ("res2_conv3", "res2_skip"):
            w *= out_scale
        sd[name + ".weight"] = w.astype(np.float32)
        sd[name + ".bias"] = np.zeros(co, np.float32)
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


def render_room_sequence(seed=2089, n_frames=48, h=480, w=640, focal=525.0, arc_deg=100.0, radius=0.9, device="cpu", pose_override=None):
    """A view-consistent synthetic mapping sequence (SURVEY.md section 8d "synthetic scene"): a 6 x 4 x 3 m box room whose walls
    carry a multi-octave value-noise texture defined in WORLD coordinates, seen by a camera that moves on an arc around the
    room centre looking outwards. Returns a dict of torch tensors on `device`:

      images  [n,1,h,w] float32 normalised as dataset.py:150-153 ((gray - 0.4) / 0.25)
      poses   [n,4,4] float32 camera -> world      depth [n,h/8,w/8] float32 camera z at the feature-map pixel centres
      focal, ppx, ppy

    Evaluated with torch ops so that it runs on the GPU box's device in milliseconds; nothing here is part of the product path.
    """
    import torch
    g = torch.Generator().manual_seed(seed)
    dev = torch.device(device)
    room = torch.tensor([6.0, 4.0, 3.0])
    centre = room / 2
    # camera centres on an arc in the horizontal (x, z) plane at mid height; y axis of the camera = world y
    ang = torch.deg2rad(torch.linspace(-arc_deg / 2, arc_deg / 2, n_frames)) + 0.3
    bob = 0.05 * torch.sin(torch.linspace(0, 6.0, n_frames))
    poses = torch.eye(4).repeat(n_frames, 1, 1)
    for i in range(n_frames):
        a = float(ang[i])
        fwd = torch.tensor([math.sin(a), 0.0, math.cos(a)])          # looking outwards
        right = torch.tensor([math.cos(a), 0.0, -math.sin(a)])
        down = torch.tensor([0.0, 1.0, 0.0])
        R = torch.stack([right, down, fwd], dim=1)                    # columns = camera axes in world
        pitch = 0.08 * math.sin(0.7 * i)
        Rx = torch.tensor([[1, 0, 0], [0, math.cos(pitch), -math.sin(pitch)], [0, math.sin(pitch), math.cos(pitch)]], dtype=torch.float32)
        poses[i, :3, :3] = R @ Rx
        poses[i, :3, 3] = centre + radius * fwd * 0.6 + torch.tensor([0.0, float(bob[i]), 0.0])
    if pose_override is not None:                                        # same room and texture, caller-chosen cameras [n,4,4]
        poses = torch.as_tensor(pose_override, dtype=torch.float32).reshape(-1, 4, 4).cpu()
        n_frames = poses.shape[0]
    # value-noise lattices: 6 walls x 4 octaves (cell sizes 0.6 / 0.2 / 0.06 / 0.02 m), bilinear lookups
    cells = (0.6, 0.2, 0.06, 0.02)
    amps = (0.30, 0.30, 0.25, 0.15)
    lat = [[torch.rand((1, 1, int(6.0 / c) + 3, int(6.0 / c) + 3), generator=g).to(dev) for c in cells] for _ in range(6)]
    poses_d = poses.to(dev)
    ppx, ppy = w / 2.0, h / 2.0

    def shade(P, wall):
        """P [...,3] world points, wall [...] in 0..5 -> gray in [0,1]."""
        out = torch.zeros(P.shape[:-1], device=dev)
        for wid in range(6):
            m = wall == wid
            if not bool(m.any()):
                continue
            axis = wid // 2
            uv = P[m][:, [a for a in range(3) if a != axis]]          # the two in-plane coordinates, metres
            val = torch.zeros(uv.shape[0], device=dev)
            for o, c in enumerate(cells):
                L = lat[wid][o]
                n = L.shape[-1]
                gxy = (uv / c + 1.0) / (n - 1) * 2 - 1                # lattice coordinates -> [-1,1]
                smp = torch.nn.functional.grid_sample(L, gxy.view(1, 1, -1, 2), mode="bilinear", align_corners=True).view(-1)
                val += amps[o] * smp
                if o == 1:
                    blob = torch.sigmoid((smp - 0.62) * 40.0)         # soft-edged "posters" at the 0.2 m scale
            out[m] = (val - 0.5) * 1.7 + 0.45 + 0.3 * blob
        return out.clamp(0, 1)

    def cast(pose, xs, ys):
        gx, gy = torch.meshgrid(xs, ys, indexing="xy")
        rays = torch.stack([(gx - ppx) / focal, (gy - ppy) / focal, torch.ones_like(gx)], dim=-1).to(dev)   # [H,W,3]
        d = rays @ pose[:3, :3].T
        c = pose[:3, 3]
        roomd = room.to(dev)
        t1 = (0.0 - c) / d
        t2 = (roomd - c) / d
        t = torch.where(d > 0, t2, t1)
        t = torch.where(d.abs() < 1e-12, torch.full_like(t, float("inf")), t)
        tmin, axis = t.min(dim=-1)
        P = c + d * tmin.unsqueeze(-1)
        side = (torch.gather(d, -1, axis.unsqueeze(-1)).squeeze(-1) > 0).long()
        return P, axis * 2 + side, tmin * rays[..., 2]                 # camera z = t * ray_z (ray_z = 1)

    images = torch.empty((n_frames, 1, h, w), device=dev)
    depth = torch.empty((n_frames, h // 8, w // 8), device=dev)
    px = torch.arange(w, dtype=torch.float32) + 0.5
    py = torch.arange(h, dtype=torch.float32) + 0.5
    cx = torch.arange(w // 8, dtype=torch.float32) * 8 + 4
    cy = torch.arange(h // 8, dtype=torch.float32) * 8 + 4
    for i in range(n_frames):
        P, wall, _ = cast(poses_d[i], px, py)
        images[i, 0] = (shade(P, wall) - 0.4) / 0.25
        _, _, z = cast(poses_d[i], cx, cy)
        depth[i] = z
    return {"images": images, "poses": poses_d, "depth": depth, "focal": float(focal), "ppx": ppx, "ppy": ppy}

"""Point-cloud extraction on the device: host mirror of
ace_vis_util.get_point_cloud_from_network (ace_vis_util.py:430-591) and of export_point_cloud.py's txt / ply writers.

    xyz, clr = get_point_cloud_from_network(network, frames, filter_depth=100, dense_cloud=False)

`network` is an acezero_amd.network.Regressor; `frames` is a sized iterable of
(image_B1HW float32 normalised, pose_inv_B44 world->camera, K_B33[, rgb_BHW3 uint8 at network input resolution]) batches --
what the reference's data loader yields per frame (batch size 1 there, any batch size here).  Scene coordinates never leave
HBM: encoder -> head -> acez_point_cloud_filter, one [N,3] array comes back at the end.  Colours are looked up on the host at
the pixel centre of every kept map pixel (nearest-neighbour sub-sampling, :566-570); without an rgb array the normalised gray
value of the network input is used.  There is no CPU fallback.
"""
import numpy as np
import torch

from . import _native as N
from .head import _ptr, _stream

PC_POINTS_MIN, PC_POINTS_MAX = 100000, 1000000   # ace_vis_util.py:447-448


def points_per_image(n_frames):
    """ace_vis_util.py:458-459."""
    return int(PC_POINTS_MIN / n_frames), int(PC_POINTS_MAX / n_frames)


def filter_scene_coordinates(scene_coords, poses_inv, intrinsics, filter_depth, dense_cloud, n_frames_total, seed=0,
                             first_frame_id=0, opengl=True, want_source=True):
    """Device tensors in, device tensors out: (xyz [N,3] f32, source [N] i32 or None, counts [B] i32, keep [B,h*w] u8).

    scene_coords [B,3,h,w] f32, poses_inv [B,4,4] or [B,3,4] world->camera, intrinsics [B,3,3]."""
    if not scene_coords.is_cuda:
        raise RuntimeError("filter_scene_coordinates needs device tensors: the point-cloud filter is a HIP kernel, there is no CPU path")
    dev = scene_coords.device
    b, _, h, w = scene_coords.shape
    sc = scene_coords.to(torch.float32).contiguous()
    pinv = poses_inv.to(dev, torch.float32)[:, :3, :].contiguous()
    K = intrinsics.to(dev, torch.float32).contiguous()
    pmin, pmax = points_per_image(int(n_frames_total))
    keep = torch.empty((b, h * w), dtype=torch.uint8, device=dev)
    counts = torch.empty((b,), dtype=torch.int32, device=dev)
    offsets = torch.empty((b + 1,), dtype=torch.int32, device=dev)
    xyz = torch.empty((b * h * w, 3), dtype=torch.float32, device=dev)
    src = torch.empty((b * h * w,), dtype=torch.int32, device=dev) if want_source else None
    N.check(N.lib().acez_point_cloud_filter(_ptr(sc), _ptr(pinv), _ptr(K), b, h, w, float(filter_depth), int(bool(dense_cloud)), pmin, pmax,
                                            int(seed), int(first_frame_id), int(bool(opengl)), _ptr(keep), _ptr(counts), _ptr(offsets),
                                            _ptr(xyz), _ptr(src) if want_source else None, _stream()))
    n = int(offsets[b].item())
    return xyz[:n], (src[:n] if want_source else None), counts, keep


def get_point_cloud_from_network(network, frames, filter_depth, dense_cloud=False, seed=0):
    """Returns (pc_xyz [N,3] float32, pc_clr [N,3] float64) in the OpenGL convention, as the reference does."""
    n_total = sum(int(batch[0].shape[0]) for batch in frames) if not hasattr(frames, "num_frames") else int(frames.num_frames)
    pc_xyz, pc_clr = [], []
    frame_id = 0
    for batch in frames:
        image, pose_inv, K = batch[0], batch[1], batch[2]
        rgb = batch[3] if len(batch) > 3 else None
        image = image.to(network.device, torch.float32)
        sc = network(image)
        b, _, h, w = sc.shape
        xyz, src, _, _ = filter_scene_coordinates(sc, pose_inv, K, filter_depth, dense_cloud, n_total, seed=seed, first_frame_id=frame_id)
        src = src.cpu().numpy().astype(np.int64)
        f, p = np.divmod(src, h * w)
        y, x = np.divmod(p, w)
        off = network.OUTPUT_SUBSAMPLE // 2
        yy = np.minimum(y * network.OUTPUT_SUBSAMPLE + off, image.shape[2] - 1)
        xx = np.minimum(x * network.OUTPUT_SUBSAMPLE + off, image.shape[3] - 1)
        if rgb is not None:
            clr = np.asarray(rgb)[f, yy, xx].astype(np.float64)             # 0..255 floats, as the reference's float64 resize keeps them (:560-562)
        else:
            gray = (image[:, 0].cpu().numpy()[f, yy, xx].astype(np.float64) * 0.25 + 0.4) * 255.0   # undo dataset.py:150-153
            clr = np.repeat(np.clip(gray, 0.0, 255.0)[:, None], 3, axis=1)
        pc_xyz.append(xyz.cpu().numpy())
        pc_clr.append(clr)
        frame_id += b
    return np.concatenate(pc_xyz, 0), np.concatenate(pc_clr, 0)


def write_point_cloud(path, pc_xyz, pc_clr):
    """export_point_cloud.py:109-125: '.txt' = 'x y z r g b' lines (colours 0-255, printed with %.0f), '.ply' = binary
    little-endian vertex list with uchar colours (what trimesh.PointCloud.export writes)."""
    path = str(path)
    xyz = np.asarray(pc_xyz, np.float32)
    clr = np.asarray(pc_clr, np.float64)
    if path.endswith(".txt"):
        with open(path, "w") as fh:
            for pt in range(xyz.shape[0]):
                fh.write(f"{xyz[pt, 0]} {xyz[pt, 1]} {xyz[pt, 2]} {clr[pt, 0]:.0f} {clr[pt, 1]:.0f} {clr[pt, 2]:.0f}\n")
    elif path.endswith(".ply"):
        rec = np.zeros(len(xyz), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1"), ("a", "u1")])
        rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
        c8 = np.clip(np.rint(clr), 0, 255).astype(np.uint8)
        rec["r"], rec["g"], rec["b"], rec["a"] = c8[:, 0], c8[:, 1], c8[:, 2], 255
        with open(path, "wb") as fh:
            fh.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                      "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\nend_header\n" % len(xyz)).encode())
            fh.write(rec.tobytes())
    else:
        raise ValueError("output file format not supported: use .txt or .ply (export_point_cloud.py:124-125)")

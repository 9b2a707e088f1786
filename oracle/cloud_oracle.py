"""CPU oracle of the point-cloud extraction filter -- TEST INFRASTRUCTURE ONLY (see oracle/head_oracle.py for the rule).

Restates the per-frame filtering of ace_vis_util.get_point_cloud_from_network (ace_vis_util.py:430-591) from the point
where the scene-coordinate map of a frame exists: reprojection error (:481-499), scene-coordinate gradient with the
reflect padding (:501-510), escalating gradient thresholds (:512-516), depth filter (:518-526), reprojection filter with
the relaxed k-th error (:528-544) and the random sub-sampling (:545-551), the OpenGL flip (:585-587).

fp32, one rounding per operation, fixed left-to-right order, no fused multiply-add: the HIP kernel
(acezero_amd/csrc/cloud_api.hip, built with -ffp-contract=off) evaluates the same expression tree, so the keep masks
agree bit for bit.  The reference's matmul order is a BLAS detail; tests/golden/make_cloud_golden.py runs the
reference function itself and tests/test_cloud_oracle.py pins this restatement on its output.

The sub-sampling draw (`torch.randperm(n) < k`, a uniformly random k-subset from torch's global CPU generator) is not
reproducible outside torch: as for RANSAC and buffer sampling, parity is defined on a counter-based stream keyed by
(seed, frame id, rank of the point); the LAW (uniform k-subset of the surviving points, k as the reference computes it)
is the reference's.
"""
import numpy as np

from .buffer_oracle import M64, smix64

F = np.float32
GRAD_THRESHOLDS = (F(0.1), F(0.5), F(1.0), F(np.inf))   # ace_vis_util.py:443
REPRO_THRESHOLD = F(1.0)                                 # :451
PC_POINTS_MIN, PC_POINTS_MAX = 100000, 1000000           # :447-448


def points_per_image(n_frames):
    """ace_vis_util.py:458-459."""
    return int(PC_POINTS_MIN / n_frames), int(PC_POINTS_MAX / n_frames)


def cloud_draw(seed, frame_id, j):
    return smix64((smix64((seed ^ ((frame_id * 0xA0761D6478BD642F) & M64)) & M64) + j) & M64) >> 32


def frame_quantities(sc, pose_inv, K):
    """sc f32 [3,h,w]; pose_inv f32 [3,4] world->camera; K f32 [3,3]. Returns err [hw], grad [hw], cam_z [hw] (all f32)."""
    sc = np.asarray(sc, F)
    P = np.asarray(pose_inv, F)
    K = np.asarray(K, F)
    _, h, w = sc.shape
    X = sc.reshape(3, -1)
    cam = [((P[i, 0] * X[0] + P[i, 1] * X[1]) + P[i, 2] * X[2]) + P[i, 3] for i in range(3)]
    px = [(K[i, 0] * cam[0] + K[i, 1] * cam[1]) + K[i, 2] * cam[2] for i in range(3)]
    z = np.maximum(px[2], F(0.1))
    ys, xs = np.divmod(np.arange(h * w), w)
    gx = F(8.0) * (xs.astype(F) + F(0.5))
    gy = F(8.0) * (ys.astype(F) + F(0.5))
    with np.errstate(all="ignore"):
        err = np.abs(px[0] / z - gx) + np.abs(px[1] / z - gy)
    # gradient: difference to the left / upper neighbour; column 0 / row 0 take the value of column 1 / row 1
    # (F.pad(..., mode='reflect') of the (w-1)- / (h-1)-long difference arrays)
    d = sc[:, :, 1:] - sc[:, :, :-1]
    g = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
    grad_x = np.concatenate([g[:, 1:2], g], axis=1)
    d = sc[:, 1:, :] - sc[:, :-1, :]
    g = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
    grad_y = np.concatenate([g[1:2, :], g], axis=0)
    grad = np.maximum(grad_x, grad_y).reshape(-1)
    return err.astype(F), grad.astype(F), cam[2].astype(F)


def filter_frame(sc, pose_inv, K, filter_depth, dense, pts_min, pts_max, seed=0, frame_id=0):
    """Keep mask (bool [h*w]) of one frame + a small dict naming the branch taken."""
    err, grad, cam_z = frame_quantities(sc, pose_inv, K)
    thresholds = (F(np.inf),) if dense else GRAD_THRESHOLDS
    repro_thr = F(np.inf) if dense else REPRO_THRESHOLD
    for ti, thr in enumerate(thresholds):
        grad_mask = grad < thr
        if int(grad_mask.sum()) > pts_min:
            break
    mask = grad_mask & (cam_z < F(filter_depth))
    info = {"grad_threshold": float(thr), "branch": "plain"}
    if int(mask.sum()) == 0:
        mask = np.ones_like(mask)
        info["all_kept"] = True
    keep = (err < repro_thr) & mask
    n_valid = int(keep.sum())
    if n_valid < pts_min:
        srt = np.sort(err[mask], kind="stable")
        relaxed = srt[min(pts_min, len(srt) - 1)]
        keep = (err < relaxed) & mask
        info["branch"] = "relaxed"
    elif n_valid > pts_max:
        keep_ratio = pts_max / n_valid
        k = int(keep_ratio * n_valid)
        idx = np.flatnonzero(keep)
        hsh = np.array([cloud_draw(seed, frame_id, j) for j in range(n_valid)], np.uint64)
        hk = np.sort(hsh)[k]
        sel = hsh < hk
        need = k - int(sel.sum())
        ties = np.flatnonzero(hsh == hk)[:need]
        sel[ties] = True
        keep = np.zeros_like(keep)
        keep[idx[sel]] = True
        info["branch"] = "subsampled"
        info["k"] = k
    return keep, info


def point_cloud(scs, poses_inv, Ks, filter_depth, dense, n_frames_total=None, seed=0, first_frame_id=0, opengl=True):
    """All frames: returns (xyz f32 [N,3], source i32 [N] = frame * hw + pixel, counts i32 [n], keep bool [n, hw])."""
    n = len(scs)
    pts_min, pts_max = points_per_image(n_frames_total or n)
    xyz, src, counts, keeps = [], [], [], []
    hw = scs[0].shape[1] * scs[0].shape[2]
    for f in range(n):
        keep, _ = filter_frame(scs[f], poses_inv[f], Ks[f], filter_depth, dense, pts_min, pts_max, seed, first_frame_id + f)
        p = np.asarray(scs[f], F).reshape(3, -1)[:, keep].T.copy()
        if opengl:
            p[:, 1] = -p[:, 1]
            p[:, 2] = -p[:, 2]
        xyz.append(p)
        src.append((f * hw + np.flatnonzero(keep)).astype(np.int32))
        counts.append(int(keep.sum()))
        keeps.append(keep)
    return np.concatenate(xyz, 0), np.concatenate(src, 0), np.array(counts, np.int32), np.stack(keeps, 0)

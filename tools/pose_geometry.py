"""Gauge-aware pose comparison for the session studies (tools/reconstruct_synth.py, tools/refit_study.py): similarity alignment of
the camera centres (scale, centre error relative to the trajectory extent), rotation error after a rotation-only alignment (the
rotation of a centre alignment is ill-determined when the centres are distorted), consecutive-frame relative rotation (gauge free)."""
import numpy as np


def similarity_fit(src, dst):
    mu_s, mu_d = src.mean(0), dst.mean(0)
    a, b = src - mu_s, dst - mu_d
    U, S, Vt = np.linalg.svd(b.T @ a / len(src))
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    return float(np.trace(np.diag(S) @ D) / (a ** 2).sum() * len(src)), R, None


def geometry(est, gt):
    """est, gt: cam->world [k,4,4] of the same frames (in sequence order)."""
    est, gt = est.astype(np.float64), gt.astype(np.float64)
    s, R, _ = similarity_fit(est[:, :3, 3], gt[:, :3, 3])
    c = s * (est[:, :3, 3] - est[:, :3, 3].mean(0)) @ R.T + gt[:, :3, 3].mean(0)
    extent = float(np.linalg.norm(gt[:, :3, 3].max(0) - gt[:, :3, 3].min(0)))
    dc = np.linalg.norm(c - gt[:, :3, 3], axis=1) / extent
    # rotation-only alignment: Q = argmin sum |Q R_est - R_gt|_F
    M = np.einsum("nij,nkj->ik", gt[:, :3, :3], est[:, :3, :3])
    U, _, Vt = np.linalg.svd(M)
    Q = U @ np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))]) @ Vt
    ang = lambda A, B: np.degrees(np.arccos(np.clip((np.einsum("nij,nij->n", A, B) - 1) / 2, -1, 1)))
    rot_abs = ang(np.einsum("ij,njk->nik", Q, est[:, :3, :3]), gt[:, :3, :3])
    rel_e = np.einsum("nji,njk->nik", est[:-1, :3, :3], est[1:, :3, :3])
    rel_g = np.einsum("nji,njk->nik", gt[:-1, :3, :3], gt[1:, :3, :3])
    # the est trajectory's extent relative to its own scene: arc radius estimate = extent / gt extent * s^-1 is the similarity scale
    return {"scale": round(s, 4), "centre_rel_median": round(float(np.median(dc)), 4), "centre_rel_p90": round(float(np.quantile(dc, 0.9)), 4),
            "rot_abs_deg_median": round(float(np.median(rot_abs)), 3), "rot_rel_deg_median": round(float(np.median(ang(rel_e, rel_g))), 4),
            "rot_rel_deg_p90": round(float(np.quantile(ang(rel_e, rel_g), 0.9)), 4)}

#!/usr/bin/env python
"""Golden vectors for the DSAC* path from the REFERENCE's own environment (OpenCV 4.4.0 + the reference's dsacstar build).

This cannot run in the build container (no OpenCV, dsacstar cannot be compiled: oracle/dsac_oracle.cpp header). Run it once
inside the reference's conda env (environment.yml: libopencv 4.4.0, py-opencv 4.4.0), after `python dsacstar/setup.py install`:

    OMP_NUM_THREADS=1 python tests/golden/make_dsac_golden.py --reference /path/to/acezero --out tests/golden/dsac_ref.npz
    OMP_NUM_THREADS=12 python tests/golden/make_dsac_golden.py --reference /path/to/acezero --out tests/golden/dsac_ref_t12.npz
                                                                       # (register_mapping.py:8 runs with 12 threads)

(--reference: the reference checkout, put on sys.path for `import dsacstar` -- the extension built in place by `python setup.py
build_ext --inplace` inside dsacstar/ works as well as an installed one. Nothing else of this repository is imported: the script needs
numpy, torch, cv2 and the reference's dsacstar only.)

and commit the .npz. tests/test_dsac_reference_golden.py then pins the oracle (and through the bit-exact GPU test the kernels) on
  * cv2.solvePnP(SOLVEPNP_P3P) on seeded minimal sets                         (dsacstar_util.h:104-112 via :185-193)
  * cv2.solvePnP(SOLVEPNP_ITERATIVE, useExtrinsicGuess=True)                   (:570-580)
  * cv2.projectPoints, incl. points at and behind the camera plane             (:199-205, :395-401)
  * cv2.Rodrigues in both directions                                           (:762)
  * whole dsacstar.forward_rgb calls (dsacstar.cpp:66-186) with the per-thread mt19937 stream of thread_rand.cpp, whose
    consumption depends on OMP_NUM_THREADS (recorded in the file) and on the compiler's libstdc++ (the oracle uses GCC >= 11's
    uniform_int_distribution; if the whole-call cases disagree but the unit cases agree, that is the first thing to check).
Every input is generated here from numpy's PCG64 with fixed seeds and stored next to the output, so the test needs nothing else."""
import argparse
import os
import sys

import numpy as np

F, CX, CY = 525.0, 320.0, 240.0
K = np.array([[F, 0, CX], [0, F, CY], [0, 0, 1]], np.float64)


def rodrigues_np(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def random_pose(rng):
    return rng.normal(size=3) * 0.6, rng.normal(size=3) * 0.5 + np.array([0.0, 0.0, 1.0])


def scene_points(rng, r, t, n, noise_px):
    R = rodrigues_np(r)
    Xc = rng.uniform(-1.5, 1.5, size=(n, 3)) + np.array([0, 0, 4.0])
    Xw = (Xc - t) @ R
    uv = np.stack([F * Xc[:, 0] / Xc[:, 2] + CX, F * Xc[:, 1] / Xc[:, 2] + CY], 1) + rng.normal(size=(n, 2)) * noise_px
    return Xw.astype(np.float32), uv.astype(np.float32)


def room_frames(rng, n_frames, h=60, w=80, sub=8, noise=0.02, outliers=0.3):
    """Scene-coordinate maps of cameras inside a 6 x 4 x 3 m box (the same construction as acezero_amd.synth, self-contained)."""
    room = np.array([6.0, 4.0, 3.0])
    xs = np.arange(w) * sub + sub // 2
    ys = np.arange(h) * sub + sub // 2
    gx, gy = np.meshgrid(xs.astype(np.float64), ys.astype(np.float64))
    ppx, ppy = w * sub / 2.0, h * sub / 2.0
    rays = np.stack([(gx - ppx) / F, (gy - ppy) / F, np.ones_like(gx)], 0).reshape(3, -1)
    out = np.zeros((n_frames, 3, h, w), np.float32)
    for i in range(n_frames):
        c = room * rng.uniform(0.3, 0.7, size=3)
        R = rodrigues_np(rng.normal(size=3) * 0.4)
        d = R @ rays
        with np.errstate(divide="ignore", invalid="ignore"):
            t1, t2 = (0.0 - c[:, None]) / d, (room[:, None] - c[:, None]) / d
        tt = np.where(d > 0, t2, t1)
        tt = np.where(np.abs(d) < 1e-12, np.inf, tt)
        pts = c[:, None] + d * tt.min(axis=0) + rng.normal(0, noise, size=(3, h * w))
        bad = rng.uniform(size=h * w) < outliers
        pts[:, bad] = rng.uniform(0, 1, size=(3, int(bad.sum()))) * room[:, None]
        out[i] = pts.reshape(3, h, w).astype(np.float32)
    return out, ppx, ppy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--frames", type=int, default=50, help="whole-frame forward_rgb calls per case")
    ap.add_argument("--reference", default=None, help="path of the reference checkout (and its dsacstar/ build directory) for `import dsacstar`")
    args = ap.parse_args()
    if args.reference:
        sys.path[:0] = [args.reference, os.path.join(args.reference, "dsacstar")]
    import cv2
    import torch
    import dsacstar

    rng = np.random.default_rng(20240924)
    out = {"opencv_version": np.array(cv2.__version__), "omp_threads": np.array(int(os.environ.get("OMP_NUM_THREADS", "0"))),
           "K": K}

    # ---- P3P on minimal sets (4th point ranks the solutions); includes duplicates and collinear sets (sampling is with replacement)
    n = 300
    obj = np.zeros((n, 4, 3), np.float32); img = np.zeros((n, 4, 2), np.float32)
    ok = np.zeros(n, np.uint8); rv = np.zeros((n, 3)); tv = np.zeros((n, 3))
    for i in range(n):
        r, t = random_pose(rng)
        o, m = scene_points(rng, r, t, 4, 0.7)
        if i % 25 == 0:
            o[1], m[1] = o[0], m[0]                      # duplicate sample
        if i % 25 == 1:
            o[2] = (o[0] + o[1]) / 2                     # collinear object points
        obj[i], img[i] = o, m
        good, rvec, tvec = cv2.solvePnP(o, m, K, None, flags=cv2.SOLVEPNP_P3P)
        ok[i] = good
        if good:
            rv[i], tv[i] = rvec.ravel(), tvec.ravel()
    out.update(p3p_obj=obj, p3p_img=img, p3p_ok=ok, p3p_rvec=rv, p3p_tvec=tv)

    # ---- ITERATIVE with an extrinsic guess
    n = 60
    objs, imgs, starts, res = [], [], np.zeros((n, 6)), np.zeros((n, 6))
    counts = np.zeros(n, np.int32)
    for i in range(n):
        r, t = random_pose(rng)
        k = int(rng.integers(5, 600))
        o, m = scene_points(rng, r, t, k, 1.5)
        r0, t0 = r + rng.normal(size=3) * 0.03, t + rng.normal(size=3) * 0.05
        rvec, tvec = r0.reshape(3, 1).copy(), t0.reshape(3, 1).copy()
        good, rvec, tvec = cv2.solvePnP(o, m, K, None, rvec, tvec, True, cv2.SOLVEPNP_ITERATIVE)
        assert good
        objs.append(o); imgs.append(m); counts[i] = k
        starts[i] = np.concatenate([r0, t0]); res[i] = np.concatenate([rvec.ravel(), tvec.ravel()])
    out.update(lm_obj=np.concatenate(objs), lm_img=np.concatenate(imgs), lm_counts=counts, lm_start=starts, lm_result=res)

    # ---- projectPoints (also z <= 0) and Rodrigues
    n = 200
    poses = np.stack([np.concatenate(random_pose(rng)) for _ in range(n)])
    pts = (rng.uniform(-3, 3, size=(n, 3))).astype(np.float32)
    pts[::10, 2] = 0.0
    uv = np.zeros((n, 2))
    for i in range(n):
        # the point is given in WORLD coordinates; choose some so that the camera-space z is exactly 0 or negative
        p, _ = cv2.projectPoints(pts[i:i + 1].reshape(1, 1, 3), poses[i, :3].copy(), poses[i, 3:].copy(), K, None)
        uv[i] = p.ravel()
    out.update(proj_pose=poses, proj_pts=pts, proj_uv=uv)
    rvs = rng.normal(size=(n, 3)) * np.concatenate([np.full(50, 1e-9), np.full(50, 0.5), np.full(50, 1.5), np.full(50, 3.14159 / np.sqrt(3))])[:, None]
    Rm = np.zeros((n, 3, 3)); back = np.zeros((n, 3))
    for i in range(n):
        Rm[i] = cv2.Rodrigues(rvs[i].copy())[0]
        back[i] = cv2.Rodrigues(Rm[i].copy())[0].ravel()
    out.update(rod_rvec=rvs, rod_R=Rm, rod_back=back)

    # ---- whole forward_rgb calls, frames in order (the generators are seeded once per process and continue across frames)
    sc, ppx, ppy = room_frames(rng, args.frames)
    cases = []
    poses4 = np.zeros((2, args.frames, 4, 4), np.float32); inl = np.zeros((2, args.frames), np.int64)
    for ci, (hyp, tries) in enumerate(((32, 16), (64, 1000000))):   # ace_zero.py:140-143,233 / register_mapping.py:64-79 defaults
        cases.append((hyp, tries))
        for i in range(args.frames):
            pose = torch.zeros(4, 4)
            inl[ci, i] = dsacstar.forward_rgb(torch.from_numpy(sc[i:i + 1]), pose, hyp, 10.0, F, ppx, ppy, 100.0, 100.0, 8, 1305, tries)
            poses4[ci, i] = pose.numpy()
    # case 2: the Mip-NeRF-360-garden map size (480 x 741 frames -> 60 x 93 coordinates), ace_zero's 32 hypotheses / 16 tries
    sc93, ppx93, ppy93 = room_frames(rng, args.frames, h=60, w=93)
    pose93 = np.zeros((args.frames, 4, 4), np.float32); inl93 = np.zeros(args.frames, np.int64)
    for i in range(args.frames):
        pose = torch.zeros(4, 4)
        inl93[i] = dsacstar.forward_rgb(torch.from_numpy(sc93[i:i + 1]), pose, 32, 10.0, F, ppx93, ppy93, 100.0, 100.0, 8, 1305, 16)
        pose93[i] = pose.numpy()
    out.update(fr_sc=sc, fr_ppx=np.array(ppx), fr_ppy=np.array(ppy), fr_cases=np.array(cases), fr_pose=poses4, fr_inliers=inl,
               fr93_sc=sc93, fr93_ppx=np.array(ppx93), fr93_ppy=np.array(ppy93), fr93_pose=pose93, fr93_inliers=inl93,
               fr_note=np.array("calls were made in this order within ONE process (the generators are seeded once and continue): case 0 "
                                "(32 hyp / 16 tries) frames 0..n-1, case 1 (64 / 1e6) frames 0..n-1, then the 60x93 frames 0..n-1 with "
                                "32 / 16; seed 1305"))
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, "OpenCV", cv2.__version__, "OMP threads", os.environ.get("OMP_NUM_THREADS"))


if __name__ == "__main__":
    sys.exit(main())

"""CPU: evidence for the DSAC* oracle that does NOT share code with it ("R parity is the same text compiled twice").

The oracle (oracle/dsac_oracle.cpp) restates OpenCV's P3P / Levenberg-Marquardt from memory and the GPU kernel shares most of
that text, so a wrong root ranking or stopping rule would sit on both sides of the bit-exact GPU test. Here the same questions
are answered by different means:
  * P3P: every real solution of the three-point problem is found by a formula-free method (1-D scan of the law-of-cosines
    system + Brent refinement, Kabsch alignment by SVD in numpy) and the oracle's answer must be the candidate with the smallest
    reprojection error of the fourth point (cv::solvePnP(SOLVEPNP_P3P) semantics, dsacstar_util.h:185-193);
  * LM: from the same start scipy.optimize.least_squares (MINPACK) on the pixel reprojection residuals, with scipy's own
    Rodrigues, reaches the same pose (dsacstar_util.h:566-581, cv::solvePnP(ITERATIVE, useExtrinsicGuess));
  * the Cholesky solver of deviation D4 against the SVD-equivalent solver OpenCV runs: identical inlier masks and poses on the
    frames of the GPU parity test;
  * the reference-style mt19937 stream (thread_rand.cpp) and the generic point-list entry that the reference-environment golden
    vectors (tests/golden/make_dsac_golden.py -> dsac_ref.npz, consumed by test_dsac_reference_golden.py) need."""
import numpy as np
import pytest
from scipy.optimize import brentq, least_squares
from scipy.spatial.transform import Rotation

from acezero_amd import synth
from oracle import dsac_oracle

F, CX, CY = 525.0, 320.0, 240.0


def _random_pose(rng):
    R = Rotation.from_rotvec(rng.normal(size=3) * 0.6).as_matrix()
    t = rng.normal(size=3) * 0.5 + np.array([0.0, 0.0, 1.0])
    return R, t


def _project(R, t, X):
    Xc = X @ R.T + t
    return np.stack([F * Xc[:, 0] / Xc[:, 2] + CX, F * Xc[:, 1] / Xc[:, 2] + CY], axis=1)


def _all_p3p_poses(obj3, img3):
    """Every pose that maps the three world points onto their three image rays, without the quartic: scan the depth of point 1."""
    rays = np.stack([(img3[:, 0] - CX) / F, (img3[:, 1] - CY) / F, np.ones(3)], axis=1)
    rays /= np.linalg.norm(rays, axis=1, keepdims=True)
    a2 = np.sum((obj3[1] - obj3[2]) ** 2)
    b2 = np.sum((obj3[0] - obj3[2]) ** 2)
    c2 = np.sum((obj3[0] - obj3[1]) ** 2)
    ca, cb, cg = rays[1] @ rays[2], rays[0] @ rays[2], rays[0] @ rays[1]
    sg2, sb2 = 1 - cg * cg, 1 - cb * cb
    s1max = min(np.sqrt(c2 / sg2), np.sqrt(b2 / sb2))
    grid = np.linspace(1e-6, s1max * (1 - 1e-9), 6000)
    sols = []
    for sg in (1.0, -1.0):
        for sb in (1.0, -1.0):
            def g(s1):
                s2 = s1 * cg + sg * np.sqrt(np.maximum(c2 - s1 * s1 * sg2, 0.0))
                s3 = s1 * cb + sb * np.sqrt(np.maximum(b2 - s1 * s1 * sb2, 0.0))
                return s2 * s2 + s3 * s3 - 2 * s2 * s3 * ca - a2, s2, s3
            v = g(grid)[0]
            for i in np.nonzero(np.sign(v[:-1]) * np.sign(v[1:]) < 0)[0]:
                s1 = brentq(lambda s: g(s)[0], grid[i], grid[i + 1], xtol=1e-15, rtol=1e-15)
                _, s2, s3 = g(s1)
                if s2 > 0 and s3 > 0:
                    sols.append((s1, float(s2), float(s3)))
    poses = []
    for s in sols:
        Q = rays * np.array(s)[:, None]          # camera-frame points
        mq, mp = Q.mean(0), obj3.mean(0)
        H = (obj3 - mp).T @ (Q - mq)
        U, _, Vt = np.linalg.svd(H)
        D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
        R = Vt.T @ D @ U.T
        t = mq - R @ mp
        if not any(np.allclose(R, R2, atol=1e-7) and np.allclose(t, t2, atol=1e-7) for R2, t2 in poses):
            poses.append((R, t))
    return poses


def test_p3p_returns_the_root_with_the_smallest_fourth_point_error():
    rng = np.random.default_rng(7)
    checked = found = 0
    for _ in range(1500):
        R, t = _random_pose(rng)
        X = rng.uniform(-1.5, 1.5, size=(4, 3)) + np.array([0, 0, 4.0])
        Xw = (X - t) @ R                               # world points whose camera coordinates are X
        uv = _project(R, t, Xw) + rng.normal(size=(4, 2)) * 0.7
        obj = Xw.astype(np.float32)
        img = uv.astype(np.float32)
        ok, pose6 = dsac_oracle.p3p(obj, img, F, CX, CY)
        cands = _all_p3p_poses(obj[:3].astype(np.float64), img[:3].astype(np.float64))
        if not ok or not cands:
            continue
        errs = [np.linalg.norm(_project(Rc, tc, obj[3:4].astype(np.float64))[0] - img[3]) for Rc, tc in cands]
        order = np.argsort(errs)
        if len(cands) > 1 and errs[order[1]] - errs[order[0]] < 1e-3:
            continue                                    # a near tie is decided by rounding, not by the algorithm
        Ro = Rotation.from_rotvec(pose6[:3]).as_matrix()
        dist = [max(np.abs(Ro - Rc).max(), np.abs(pose6[3:] - tc).max() / max(1.0, np.abs(tc).max())) for Rc, tc in cands]
        if min(dist) > 1e-4:
            continue                                    # the scan missed this root (tangency): nothing to compare with
        found += 1
        assert int(np.argmin(dist)) == int(order[0]), (errs, dist)
        checked += 1
    assert checked >= 1000, (checked, found)


def test_lm_refinement_agrees_with_minpack_from_the_same_start():
    rng = np.random.default_rng(11)
    worst = 0.0
    for case in range(40):
        R, t = _random_pose(rng)
        n = int(rng.integers(12, 400))
        X = rng.uniform(-2, 2, size=(n, 3)) + np.array([0, 0, 5.0])
        Xw = ((X - t) @ R).astype(np.float32)
        uv = (_project(R, t, Xw.astype(np.float64)) + rng.normal(size=(n, 2)) * 1.5).astype(np.float32)
        r0 = Rotation.from_matrix(R).as_rotvec() + rng.normal(size=3) * 0.03
        t0 = t + rng.normal(size=3) * 0.05
        start = np.concatenate([r0, t0])
        got = dsac_oracle.pnp_iterative_pts(Xw, uv, F, CX, CY, start)

        def res(p):
            return (_project(Rotation.from_rotvec(p[:3]).as_matrix(), p[3:], Xw.astype(np.float64)) - uv).ravel()
        ref = least_squares(res, start, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15).x
        worst = max(worst, float(np.abs(got - ref).max()))
        # a stationary point of the pixel reprojection error: same cost as MINPACK's optimum, tiny gradient
        c_got, c_ref = np.sum(res(got) ** 2), np.sum(res(ref) ** 2)
        assert c_got <= c_ref * (1 + 1e-9) + 1e-12
        np.testing.assert_allclose(got, ref, atol=1e-6, rtol=0)
    assert worst < 1e-6


def _frames():
    out = []
    for seed, kw in ((3, {}), (5, dict(outlier_ratio=0.6)), (9, dict(noise_sigma=0.05)), (1305, {})):
        fr = synth.make_registration_frames(seed=seed, n_frames=4, **kw)
        out += [(fr, i) for i in range(4)]
    return out


def test_svd_and_cholesky_solvers_give_identical_masks_and_poses():
    """Deviation D4 (Cholesky instead of cv::solve(DECOMP_SVD)) must not change a single inlier decision."""
    try:
        for hyps in (32, 64):
            for fr, i in _frames():
                res = {}
                for mode in ("cholesky", "svd"):
                    dsac_oracle.set_options(solver=mode)
                    res[mode] = dsac_oracle.forward_rgb(fr["scene_coords"][i], hyps, 10.0, fr["focal"], fr["ppx"], fr["ppy"], 100.0, 100.0, 8,
                                                        1305, i, 16)
                a, b = res["cholesky"], res["svd"]
                assert a["inliers"] == b["inliers"] and np.array_equal(a["mask"], b["mask"]) and a["best"] == b["best"]
                # the LM iteration stops on a relative parameter change of FLT_EPSILON: the two solvers' poses agree to that level
                np.testing.assert_allclose(a["refined"], b["refined"], rtol=0, atol=2e-7)
                np.testing.assert_allclose(a["pose"], b["pose"], rtol=0, atol=1e-6)
    finally:
        dsac_oracle.set_options()


def test_reference_style_mt19937_stream():
    fr = synth.make_registration_frames(seed=3, n_frames=2)
    try:
        def run(threads, frames=(0, 1)):
            dsac_oracle.set_options(rng="mt19937", rng_threads=threads)   # (re-arms the one-time seeding, like a fresh process)
            return [dsac_oracle.forward_rgb(fr["scene_coords"][i], 32, 10.0, fr["focal"], fr["ppx"], fr["ppy"], 100.0, 100.0, 8, 1305, i, 16)
                    for i in frames]
        a, b = run(1), run(1)
        assert all(np.array_equal(x["hyp_poses"], y["hyp_poses"]) for x, y in zip(a, b)), "same seed, same thread count: same stream"
        c = run(12)                                                        # register_mapping.py:8 OMP_NUM_THREADS = 12
        assert not np.array_equal(a[0]["hyp_poses"], c[0]["hyp_poses"]), "the stream depends on the thread count (SURVEY 8c)"
        # thread 0's first hypotheses are drawn from mt19937(seed + 0) in both runs
        assert np.array_equal(a[0]["hyp_poses"][0], c[0]["hyp_poses"][0])
        # the generators are seeded once per process: a second frame continues the streams, frame order matters
        d = run(1, frames=(1, 0))
        assert not np.array_equal(a[1]["hyp_poses"], d[0]["hyp_poses"])
        for r in a + c:
            assert r["inliers"] > 2000                                    # still registers the frame
        # first draws = std::mt19937(1305) through uniform_int_distribution<int>(0, 79) / (0, 59) of this libstdc++: the first
        # hypothesis must have been sampled at these four cells (x first, then y)
        import subprocess, tempfile, os, textwrap
        src = textwrap.dedent("""
            #include <random>
            #include <cstdio>
            int main() { std::mt19937 g; g.seed(1305u); for (int j = 0; j < 4; ++j) { std::uniform_int_distribution<int> dx(0, 79), dy(0, 59);
              int x = dx(g); int y = dy(g); printf("%d %d\\n", x, y); } }""")
        with tempfile.TemporaryDirectory() as d_:
            open(os.path.join(d_, "m.cpp"), "w").write(src)
            subprocess.run(["g++", "-O1", "-o", os.path.join(d_, "m"), os.path.join(d_, "m.cpp")], check=True)
            cells = np.array(subprocess.run([os.path.join(d_, "m")], capture_output=True, text=True, check=True).stdout.split(), int).reshape(4, 2)
        sc = fr["scene_coords"][0]
        obj = np.stack([sc[:, y, x] for x, y in cells]).astype(np.float32)
        img = np.stack([[x * 8 + 4, y * 8 + 4] for x, y in cells]).astype(np.float32)
        ok, pose6 = dsac_oracle.p3p(obj, img, fr["focal"], fr["ppx"], fr["ppy"])
        if ok:   # (if the first try was rejected the hypothesis comes from a later try)
            uv = dsac_oracle.project(pose6, fr["focal"], fr["ppx"], fr["ppy"], obj)
            if np.all(np.linalg.norm(uv - img, axis=1) < 10.0):
                np.testing.assert_allclose(a[0]["hyp_poses"][0], pose6, rtol=0, atol=0)
    finally:
        dsac_oracle.set_options()


def test_scores_selection_and_mask_recomputed_in_numpy():
    """getHypScores / softMax + argmax / the final inlier map (dsacstar_util.h:316-343, 356-446, 684-752, 522-597) recomputed with
    numpy from the oracle's OWN hypothesis poses and refined pose: soft inlier score = alpha / (W H) * sum_px (1 - sigmoid(beta (e -
    tau))), beta = 5 / tau, e = min(|reprojection error|, maxReproj); selected = argmax; mask = (e(refined) < tau), count = its sum.
    Different code (vectorised float64 numpy, scipy's Rodrigues), same numbers up to the float32 roundings the reference applies."""
    thr, alpha, maxr, sub = 10.0, 100.0, 100.0, 8
    fr = synth.make_registration_frames(seed=123, n_frames=3)
    H, W = fr["scene_coords"].shape[2:]
    xs = (np.arange(W) * sub + sub // 2).astype(np.float64)
    ys = (np.arange(H) * sub + sub // 2).astype(np.float64)
    gx, gy = np.meshgrid(xs, ys)                                         # [H, W]
    for i in range(3):
        sc = fr["scene_coords"][i]
        r = dsac_oracle.forward_rgb(sc, 32, thr, fr["focal"], fr["ppx"], fr["ppy"], alpha, maxr, sub, 1305, i, 16)
        X = sc.reshape(3, -1).T.astype(np.float64)

        def errors(pose6):
            Rm = Rotation.from_rotvec(pose6[:3]).as_matrix()
            Xc = X @ Rm.T + pose6[3:]
            z = np.where(Xc[:, 2] != 0, Xc[:, 2], 1.0)
            u = fr["focal"] * Xc[:, 0] / z + fr["ppx"]
            v = fr["focal"] * Xc[:, 1] / z + fr["ppy"]
            e = np.hypot(gx.ravel() - u, gy.ravel() - v)
            return np.minimum(e, maxr).reshape(H, W)

        beta = 5.0 / thr
        scores = np.array([alpha / (W * H) * np.sum(1.0 - 1.0 / (1.0 + np.exp(-beta * (errors(p) - thr)))) for p in r["hyp_poses"]])
        np.testing.assert_allclose(scores, r["scores"], rtol=2e-5, atol=1e-6)   # the reference rounds every error to float32
        assert int(np.argmax(scores)) == r["best"]
        # refineHyp (:522-597) returns the inlier map that PRODUCED the final pose (the set of the last accepted round), and stops when a
        # round's count does not exceed the best so far: the final pose's own inliers are therefore no more than the returned count,
        # and the two sets differ by the few pixels that crossed the threshold in the last round
        e = errors(r["refined"])
        near = np.abs(e - thr) < 1e-3                                         # float32 rounding of the error decides these
        mask = r["mask"].astype(bool)
        assert int(mask.sum()) == r["inliers"]
        assert int((e < thr).sum()) <= r["inliers"] + int(near.sum())
        assert int(((e < thr) ^ mask).sum()) <= max(5, r["inliers"] // 25)                # a few percent of the set
        # the returned pose is the inverse of the refined (rvec, tvec): camera -> world
        T = np.eye(4)
        T[:3, :3] = Rotation.from_rotvec(r["refined"][:3]).as_matrix()
        T[:3, 3] = r["refined"][3:]
        np.testing.assert_allclose(r["pose"], np.linalg.inv(T), atol=2e-6)

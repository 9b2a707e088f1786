"""CPU: known-answer and property tests that anchor the DSAC* oracle (the reference has no tests or golden
vectors for this path and its extension cannot be built here -- see oracle/dsac_oracle.cpp header)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from acezero_amd import synth
from oracle import dsac_oracle as O

F, PX, PY = 525.0, 320.0, 240.0


def _proj(R, t, X):
    Xc = (R @ X.T).T + t
    return np.stack([F * Xc[:, 0] / Xc[:, 2] + PX, F * Xc[:, 1] / Xc[:, 2] + PY], 1)


def test_rodrigues_roundtrip_and_jacobian():
    rng = np.random.default_rng(0)
    for i in range(100):
        r = rng.normal(0, 1.2, 3)
        R, J = O.rodrigues(r)
        np.testing.assert_allclose(R, Rot.from_rotvec(r).as_matrix(), atol=1e-14)
        th = np.linalg.norm(r)
        if th < np.pi - 1e-3:
            np.testing.assert_allclose(O.rodrigues_inv(R), r, atol=1e-10)
        eps = 1e-6
        for k in range(3):
            d = np.zeros(3); d[k] = eps
            num = (O.rodrigues(r + d)[0] - O.rodrigues(r - d)[0]).reshape(9) / (2 * eps)
            np.testing.assert_allclose(J[k], num, atol=1e-8)
    R0, J0 = O.rodrigues(np.zeros(3))
    assert np.array_equal(R0, np.eye(3))
    assert J0[0, 5] == -1 and J0[0, 7] == 1 and J0[1, 2] == 1 and J0[1, 6] == -1 and J0[2, 1] == -1 and J0[2, 3] == 1


def test_projection_jacobian_matches_finite_differences():
    rng = np.random.default_rng(1)
    pose = np.concatenate([rng.normal(0, 0.5, 3), [0.1, -0.2, 4.0]])
    xyz = rng.uniform(-1, 1, (20, 3)).astype(np.float32)
    uv, J = O.project(pose, F, PX, PY, xyz, jac=True)
    R = Rot.from_rotvec(pose[:3]).as_matrix()
    np.testing.assert_allclose(uv, _proj(R, pose[3:], xyz.astype(np.float64)), atol=1e-9)
    for k in range(6):
        d = np.zeros(6); d[k] = 1e-6
        num = (O.project(pose + d, F, PX, PY, xyz) - O.project(pose - d, F, PX, PY, xyz)) / 2e-6
        np.testing.assert_allclose(J[:, :, k], num, rtol=1e-5, atol=1e-5)


def test_quartic_solver_known_roots():
    rng = np.random.default_rng(2)
    for _ in range(200):
        roots = np.sort(rng.uniform(-3, 3, 4))
        c = np.poly(roots) * rng.uniform(0.5, 2)
        got = np.sort(O.solve_deg4(c))
        assert got.size == 4
        np.testing.assert_allclose(got, roots, atol=2e-5)
    # two real + two complex roots
    c = np.poly([1.0, 2.0, 0.5 + 1j, 0.5 - 1j]).real
    np.testing.assert_allclose(np.sort(O.solve_deg4(c)), [1.0, 2.0], atol=1e-8)
    assert O.solve_deg4(np.poly([1j, -1j, 2j, -2j]).real).size == 0


def test_p3p_known_answer_and_degenerate():
    rng = np.random.default_rng(3)
    good = 0
    for trial in range(300):
        R = Rot.random(random_state=trial).as_matrix()
        t = rng.uniform(-1, 1, 3) + [0, 0, 5.0]
        X = rng.uniform(-1.5, 1.5, (4, 3)).astype(np.float32)
        uv = _proj(R, t, X.astype(np.float64)).astype(np.float32)
        ok, pose = O.p3p(X, uv, F, PX, PY)
        if ok and np.abs(O.rodrigues(pose[:3])[0] - R).max() < 2e-3 and np.abs(pose[3:] - t).max() < 5e-3:
            good += 1
    assert good >= 285
    X = rng.uniform(-1, 1, (4, 3)).astype(np.float32)
    uv = _proj(np.eye(3), np.array([0, 0, 4.0]), X.astype(np.float64)).astype(np.float32)
    Xd, uvd = X.copy(), uv.copy()
    Xd[1], uvd[1] = Xd[0], uvd[0]      # duplicate sample (sampling is with replacement, dsacstar_util.h:168-183)
    ok, pose = O.p3p(Xd, uvd, F, PX, PY)
    assert not ok and np.all(pose == 0)  # "PnP failed" is not an error: zero pose (dsacstar_util.h:104-117)
    Xc = X.copy(); Xc[2] = Xc[0] + 2 * (Xc[1] - Xc[0])  # collinear triangle
    ok, pose = O.p3p(Xc, _proj(np.eye(3), np.array([0, 0, 4.0]), Xc.astype(np.float64)).astype(np.float32), F, PX, PY)
    assert np.all(np.isfinite(pose))


def test_sym6_solve_and_inv4x4():
    rng = np.random.default_rng(4)
    for _ in range(50):
        J = rng.normal(0, 1, (40, 6))
        A, b = J.T @ J, rng.normal(0, 1, 6)
        np.testing.assert_allclose(O.solve_sym6(A, b), np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
        T = np.eye(4); T[:3, :3] = Rot.random(random_state=7).as_matrix(); T[:3, 3] = rng.normal(0, 2, 3)
        ok, Ti = O.inv4x4(T)
        assert ok
        np.testing.assert_allclose(Ti, np.linalg.inv(T), atol=1e-14)
    J = rng.normal(0, 1, (40, 6)); J[:, 5] = J[:, 4]           # rank deficient -> pseudo-inverse (SVD semantics)
    A, b = J.T @ J, J.T @ rng.normal(0, 1, 40)
    np.testing.assert_allclose(O.solve_sym6(A, b), np.linalg.lstsq(A, b, rcond=None)[0], atol=1e-8)
    assert not O.inv4x4(np.zeros((4, 4)))[0]


def test_exact_correspondences_recover_ground_truth():
    fr = synth.make_registration_frames(seed=5, n_frames=3, noise_sigma=0.0, outlier_ratio=0.0)
    for i in range(3):
        r = O.forward_rgb(fr["scene_coords"][i], 32, 10.0, fr["focal"], fr["ppx"], fr["ppy"], 100.0, 100.0, 8, 1305, i, 16)
        assert r["inliers"] == 4800 and r["mask"].all()
        np.testing.assert_allclose(r["pose"], fr["poses"][i], atol=2e-5)


def test_noisy_frames_properties():
    fr = synth.make_registration_frames(seed=6, n_frames=4, outlier_ratio=0.5)
    for i in range(4):
        sc = fr["scene_coords"][i]
        r = O.forward_rgb(sc, 64, 10.0, fr["focal"], fr["ppx"], fr["ppy"], 100.0, 100.0, 8, 1305, i, 16)
        assert r["inliers"] == int(r["mask"].sum())
        assert 1000 < r["inliers"] < 3200
        np.testing.assert_allclose(r["pose"][:3, 3], fr["poses"][i][:3, 3], atol=0.02)
        assert r["best"] == int(np.argmax(r["scores"]))                      # argmax, first maximum wins
        # determinism + independence of the frame from call order (counter-based stream, deviation D1)
        r2 = O.forward_rgb(sc, 64, 10.0, fr["focal"], fr["ppx"], fr["ppy"], 100.0, 100.0, 8, 1305, i, 16)
        assert np.array_equal(r2["mask"], r["mask"]) and np.array_equal(r2["pose"], r["pose"])
        r3 = O.forward_rgb(sc, 64, 10.0, fr["focal"], fr["ppx"], fr["ppy"], 100.0, 100.0, 8, 1305, i + 100, 16)
        assert not np.array_equal(r3["hyp_poses"], r["hyp_poses"])
        # accessor strides are honoured (dsacstar.cpp:83-84)
        big = np.zeros((3, 60, 2 * 80), np.float32); big[:, :, ::2] = sc
        r4 = O.forward_rgb(big[:, :, ::2], 64, 10.0, fr["focal"], fr["ppx"], fr["ppy"], 100.0, 100.0, 8, 1305, i, 16)
        assert np.array_equal(r4["mask"], r["mask"])


def test_all_outliers_gives_zero_inliers_not_an_error():
    rng = np.random.default_rng(7)
    sc = rng.uniform(0, 5, (3, 60, 80)).astype(np.float32)
    r = O.forward_rgb(sc, 32, 10.0, F, PX, PY, 100.0, 100.0, 8, 1305, 0, 16)
    assert r["inliers"] < 50 and np.all(np.isfinite(r["pose"]))
    sc0 = np.zeros((3, 60, 80), np.float32)                                    # every P3P degenerate -> zero poses
    r = O.forward_rgb(sc0, 8, 10.0, F, PX, PY, 100.0, 100.0, 8, 1305, 0, 4)
    assert np.all(r["hyp_poses"] == 0)


def test_lm_refinement_converges_from_perturbed_pose():
    fr = synth.make_registration_frames(seed=8, n_frames=1, noise_sigma=0.0, outlier_ratio=0.0)
    sc = fr["scene_coords"][0]
    Tinv = np.linalg.inv(fr["poses"][0])
    gt = np.concatenate([Rot.from_matrix(Tinv[:3, :3]).as_rotvec(), Tinv[:3, 3]])
    start = gt + np.array([0.02, -0.01, 0.015, 0.05, -0.04, 0.03])
    flags = np.ones((60, 80), np.uint8); flags[::3, ::2] = 0
    out = O.pnp_iterative(sc, 8, fr["focal"], fr["ppx"], fr["ppy"], flags, start)
    np.testing.assert_allclose(out, gt, atol=1e-5)


def test_normal_equation_solver_cholesky_and_fallback():
    """D4 of oracle/dsac_oracle.cpp: Cholesky on the damped normal equations, eigen pseudo-inverse when a pivot is not safely positive."""
    rng = np.random.default_rng(12)
    for trial in range(20):
        J = rng.normal(size=(40, 6)) * np.array([300.0, 300.0, 300.0, 50.0, 50.0, 80.0])     # rotation / translation columns of a PnP Jacobian
        A = J.T @ J
        A[np.diag_indices(6)] *= 1 + 10.0 ** rng.integers(-16, 3)
        b = J.T @ rng.normal(size=40)
        x = O.solve_normal6(A, b)
        ref = np.linalg.solve(A, b)
        assert np.allclose(x, ref, rtol=1e-8, atol=1e-12 * np.abs(ref).max())
        assert np.allclose(x, O.solve_sym6(A, b), rtol=1e-7, atol=1e-11 * np.abs(ref).max())  # the two solvers agree on SPD systems
    # rank-deficient: the fallback is the pseudo-inverse solution, bit for bit
    J = rng.normal(size=(40, 6))
    J[:, 5] = J[:, 4]
    A = J.T @ J
    b = J.T @ rng.normal(size=40)
    x = O.solve_normal6(A, b)
    assert np.array_equal(x.view(np.uint64), O.solve_sym6(A, b).view(np.uint64))
    assert np.allclose(x, np.linalg.pinv(A) @ b, rtol=1e-6, atol=1e-9)

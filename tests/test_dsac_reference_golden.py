"""CPU: the DSAC* oracle against golden vectors produced by the REFERENCE's own stack (OpenCV 4.4.0 + its dsacstar build).

The vectors cannot be produced in this container (no OpenCV; tests/golden/make_dsac_golden.py explains how to generate them in
the reference's conda env). Until tests/golden/dsac_ref*.npz is committed these tests SKIP, and R parity stays "unpinned"
(oracle/dsac_oracle.cpp header); the independent checks live in tests/test_dsac_independent.py."""
import glob
import os

import numpy as np
import pytest

from oracle import dsac_oracle

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dsac_ref*.npz")))
pytestmark = pytest.mark.skipif(not FILES, reason="R PARITY UNPINNED: no tests/golden/dsac_ref*.npz. One command in the reference's conda environment "
                                                  "(OpenCV 4.4.0, dsacstar built) pins it: OMP_NUM_THREADS=1 python tests/golden/make_dsac_golden.py "
                                                  "--reference /path/to/acezero --out tests/golden/dsac_ref.npz  (then commit the .npz)")


@pytest.fixture(params=FILES or [None])
def ref(request):
    return np.load(request.param, allow_pickle=False)


def _cam(ref):
    K = ref["K"]
    return float(K[0, 0]), float(K[0, 2]), float(K[1, 2])


def test_p3p_matches_cv_solvepnp(ref):
    f, cx, cy = _cam(ref)
    for obj, img, ok, rv, tv in zip(ref["p3p_obj"], ref["p3p_img"], ref["p3p_ok"], ref["p3p_rvec"], ref["p3p_tvec"]):
        got_ok, pose = dsac_oracle.p3p(obj, img, f, cx, cy)
        assert got_ok == bool(ok)
        if ok:
            np.testing.assert_allclose(pose, np.concatenate([rv, tv]), rtol=0, atol=1e-6)


def test_iterative_with_guess_matches_cv_solvepnp(ref):
    f, cx, cy = _cam(ref)
    o = 0
    for k, start, want in zip(ref["lm_counts"], ref["lm_start"], ref["lm_result"]):
        got = dsac_oracle.pnp_iterative_pts(ref["lm_obj"][o:o + k], ref["lm_img"][o:o + k], f, cx, cy, start)
        o += int(k)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)


def test_project_points_and_rodrigues_match_opencv(ref):
    f, cx, cy = _cam(ref)
    for pose, pt, uv in zip(ref["proj_pose"], ref["proj_pts"], ref["proj_uv"]):
        np.testing.assert_allclose(dsac_oracle.project(pose, f, cx, cy, pt[None])[0], uv, rtol=1e-12, atol=1e-9)
    for r, R, back in zip(ref["rod_rvec"], ref["rod_R"], ref["rod_back"]):
        np.testing.assert_allclose(dsac_oracle.rodrigues(r)[0], R, rtol=0, atol=1e-14)
        np.testing.assert_allclose(dsac_oracle.rodrigues_inv(R), back, rtol=0, atol=1e-9)


def test_whole_forward_rgb_calls_match_the_reference_binary(ref):
    """Reference-style RNG (per-thread mt19937, seeded once, continuing across calls) + the SVD-equivalent solver OpenCV runs."""
    f, _, _ = _cam(ref)
    threads = int(ref["omp_threads"]) or 1
    try:
        dsac_oracle.set_options(solver="svd", rng="mt19937", rng_threads=threads)
        for ci, (hyp, tries) in enumerate(ref["fr_cases"]):
            for i, sc in enumerate(ref["fr_sc"]):
                r = dsac_oracle.forward_rgb(sc, int(hyp), 10.0, f, float(ref["fr_ppx"]), float(ref["fr_ppy"]), 100.0, 100.0, 8, 1305, i,
                                            int(tries))
                assert r["inliers"] == int(ref["fr_inliers"][ci, i]), (ci, i)
                np.testing.assert_allclose(r["pose"], ref["fr_pose"][ci, i], rtol=0, atol=1e-4)
        if "fr93_sc" in ref.files:   # the 60 x 93 maps (garden-sized frames) came last in the generating process
            for i, sc in enumerate(ref["fr93_sc"]):
                r = dsac_oracle.forward_rgb(sc, 32, 10.0, f, float(ref["fr93_ppx"]), float(ref["fr93_ppy"]), 100.0, 100.0, 8, 1305, i, 16)
                assert r["inliers"] == int(ref["fr93_inliers"][i]), ("60x93", i)
                np.testing.assert_allclose(r["pose"], ref["fr93_pose"][i], rtol=0, atol=1e-4)
    finally:
        dsac_oracle.set_options()

"""CPU: the host instantiation of the product's __host__ __device__ geometry (ransac_math.h) is bit-identical to
the oracle on random and degenerate inputs. This is the pre-GPU gate for the bit-exact RANSAC parity."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import dsac_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def probe():
    so = os.path.join(HERE, "_build", "libhostmath_probe.so")
    src = os.path.join(HERE, "hostmath_probe.hip")
    deps = [src] + [os.path.join(ROOT, "acezero_amd", "csrc", f) for f in ("ransac_math.h", "det_math.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                        "-o", so, src], check=True)
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def test_det_math_bitexact(probe):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-30, 30, 20000), rng.uniform(-1, 1, 20000), [0.0, -0.0, 1.0, -1.0, 1e-300, 709.5, -745.5, np.inf, -np.inf, np.nan, 1e10]])
    ref = O.det_math(x)
    outs = [np.zeros_like(x) for _ in range(5)]
    probe.probe_det_math(_p(x), x.size, *[_p(o) for o in outs])
    for name, o in zip(("sin", "cos", "acos", "exp", "cbrt"), outs):
        assert np.array_equal(o.view(np.uint64), ref[name].view(np.uint64)), name
    fin = np.isfinite(x) & (np.abs(x) < 1e9)
    np.testing.assert_allclose(ref["sin"][fin], np.sin(x[fin]), atol=3e-16)
    np.testing.assert_allclose(ref["cos"][fin], np.cos(x[fin]), atol=3e-16)


def test_p3p_bitexact_random_and_degenerate(probe):
    rng = np.random.default_rng(1)
    probe.probe_p3p.restype = C.c_int
    n_ok = 0
    for trial in range(3000):
        obj = rng.uniform(-3, 3, (4, 3)).astype(np.float32)
        img = np.stack([rng.integers(0, 80, 4) * 8 + 4, rng.integers(0, 60, 4) * 8 + 4], 1).astype(np.float32)
        if trial % 10 == 0:  # duplicates: sampling is with replacement (dsacstar_util.h:168-183)
            obj[1] = obj[0]; img[1] = img[0]
        if trial % 17 == 0:
            obj[2] = obj[0] + 2 * (obj[1] - obj[0])  # collinear
        if trial % 3 == 0:  # consistent geometry so that solutions exist
            t = np.array([0, 0, 5.0]); Xc = obj.astype(np.float64) + t
            img = np.stack([525 * Xc[:, 0] / Xc[:, 2] + 320, 525 * Xc[:, 1] / Xc[:, 2] + 240], 1).astype(np.float32)
        ok_o, pose_o = O.p3p(obj, img, 525.0, 320.0, 240.0)
        pose = np.zeros(6)
        ok = probe.probe_p3p(_p(obj), _p(img), C.c_float(525.0), C.c_float(320.0), C.c_float(240.0), _p(pose))
        assert bool(ok) == ok_o
        assert np.array_equal(pose.view(np.uint64), pose_o.view(np.uint64)), (trial, pose, pose_o)
        n_ok += int(ok_o)
    assert n_ok > 500


def test_rodrigues_project_solve_bitexact(probe):
    rng = np.random.default_rng(2)
    for trial in range(300):
        r = rng.normal(0, 1.5, 3) if trial else np.zeros(3)
        R_o, J_o = O.rodrigues(r)
        R, J = np.zeros(9), np.zeros(27)
        probe.probe_rodrigues(_p(r), _p(R), _p(J))
        assert np.array_equal(R.view(np.uint64), R_o.reshape(9).view(np.uint64))
        assert np.array_equal(J.view(np.uint64), J_o.reshape(27).view(np.uint64))
        r2 = np.zeros(3)
        probe.probe_rodrigues_inv(_p(np.ascontiguousarray(R)), _p(r2))
        assert np.array_equal(r2.view(np.uint64), O.rodrigues_inv(R_o).view(np.uint64))
        pose = np.concatenate([r, rng.normal(0, 1, 3) + [0, 0, 4]])
        xyz = rng.uniform(-2, 2, (50, 3)).astype(np.float32)
        uv_o, Jp_o = O.project(pose, 525.0, 320.0, 240.0, xyz, jac=True)
        uv, Jp = np.zeros((50, 2)), np.zeros((50, 2, 6))
        probe.probe_project(_p(pose), C.c_float(525.0), C.c_float(320.0), C.c_float(240.0), _p(xyz), 50, _p(uv), _p(Jp))
        assert np.array_equal(uv.view(np.uint64), uv_o.view(np.uint64))
        assert np.array_equal(Jp.view(np.uint64), Jp_o.view(np.uint64))
        Jm = Jp_o.reshape(-1, 6)
        A = Jm.T @ Jm
        b = rng.normal(0, 1, 6)
        x = np.zeros(6)
        probe.probe_solve_sym6(_p(np.ascontiguousarray(A)), _p(b), _p(x))
        assert np.array_equal(x.view(np.uint64), O.solve_sym6(A, b).view(np.uint64))
        Ad = A.copy()
        Ad[np.diag_indices(6)] *= 1 + 10.0 ** rng.integers(-16, 3)        # the damped system of an LM step
        probe.probe_solve_normal6(_p(np.ascontiguousarray(Ad)), _p(b), _p(x))
        assert np.array_equal(x.view(np.uint64), O.solve_normal6(Ad, b).view(np.uint64))


def test_both_det_math_copies_are_generator_output():
    """oracle/det_math.h and acezero_amd/csrc/det_math.h are not shared by copy-and-hope: both must be byte-for-byte what
    tools/gen_det_header.py emits from the mpmath fits of tools/gen_det_math.py."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_det_header.py"), "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr

"""ctypes loader of the CPU DSAC* oracle (oracle/dsac_oracle.cpp).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libdsac_oracle.so")
_lib = None


def build(force=False):
    src = [os.path.join(HERE, f) for f in ("dsac_oracle.cpp", "det_math.h", "Makefile")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in src):
        subprocess.run(["make", "-C", HERE, "-s"] + (["-B"] if force else []), check=True)
    return SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_forward_rgb.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def forward_rgb(sc, hyps, thr, focal, ppx, ppy, alpha, max_reproj, sub, seed, frame_id, max_tries, max_ref_steps=100):
    """sc: float32 [3,H,W] (any strides). Returns dict(pose [4,4] f32, inliers, mask [H,W] u8, hyp_poses, scores, best, refined)."""
    assert sc.dtype == np.float32 and sc.ndim == 3 and sc.shape[0] == 3
    H, W = sc.shape[1], sc.shape[2]
    st = [s // 4 for s in sc.strides]
    pose = np.zeros(16, np.float32)
    inl = C.c_int(0)
    mask = np.zeros((H, W), np.uint8)
    hp = np.zeros((hyps, 6), np.float64)
    scs = np.zeros(hyps, np.float64)
    best = C.c_int(0)
    ref = np.zeros(6, np.float64)
    lib().oracle_forward_rgb(_p(sc), C.c_int64(st[0]), C.c_int64(st[1]), C.c_int64(st[2]), H, W, int(hyps), C.c_float(thr),
                             C.c_float(focal), C.c_float(ppx), C.c_float(ppy), C.c_float(alpha), C.c_float(max_reproj), int(sub),
                             C.c_uint64(seed), C.c_uint64(frame_id), int(max_tries), int(max_ref_steps), _p(pose), C.byref(inl),
                             _p(mask), _p(hp), _p(scs), C.byref(best), _p(ref))
    return {"pose": pose.reshape(4, 4), "inliers": inl.value, "mask": mask, "hyp_poses": hp, "scores": scs, "best": best.value,
            "refined": ref}


def p3p(obj, img, focal, ppx, ppy):
    obj = np.ascontiguousarray(obj, np.float32).reshape(4, 3)
    img = np.ascontiguousarray(img, np.float32).reshape(4, 2)
    pose = np.zeros(6)
    lib().oracle_p3p.restype = C.c_int
    ok = lib().oracle_p3p(_p(obj), _p(img), C.c_float(focal), C.c_float(ppx), C.c_float(ppy), _p(pose))
    return bool(ok), pose


def rodrigues(r):
    r = np.ascontiguousarray(r, np.float64)
    R, J = np.zeros(9), np.zeros(27)
    lib().oracle_rodrigues(_p(r), _p(R), _p(J))
    return R.reshape(3, 3), J.reshape(3, 9)


def rodrigues_inv(R):
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    r = np.zeros(3)
    lib().oracle_rodrigues_inv(_p(R), _p(r))
    return r


def project(pose6, focal, ppx, ppy, xyz, jac=False):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    n = xyz.shape[0]
    uv = np.zeros((n, 2))
    J = np.zeros((n, 2, 6)) if jac else None
    lib().oracle_project(_p(np.ascontiguousarray(pose6, np.float64)), C.c_float(focal), C.c_float(ppx), C.c_float(ppy), _p(xyz), n,
                         _p(uv), _p(J))
    return (uv, J) if jac else uv


def solve_deg4(c5):
    roots = np.zeros(4)
    lib().oracle_solve_deg4.restype = C.c_int
    n = lib().oracle_solve_deg4(_p(np.ascontiguousarray(c5, np.float64)), _p(roots))
    return roots[:n]


def solve_sym6(A, b):
    x = np.zeros(6)
    lib().oracle_solve_sym6(_p(np.ascontiguousarray(A, np.float64)), _p(np.ascontiguousarray(b, np.float64)), _p(x))
    return x


def solve_normal6(A, b):
    x = np.zeros(6)
    lib().oracle_solve_normal6(_p(np.ascontiguousarray(A, np.float64)), _p(np.ascontiguousarray(b, np.float64)), _p(x))
    return x


def inv4x4(A):
    out = np.zeros(16)
    lib().oracle_inv4x4.restype = C.c_int
    ok = lib().oracle_inv4x4(_p(np.ascontiguousarray(A, np.float64).reshape(16)), _p(out))
    return bool(ok), out.reshape(4, 4)


def det_math(x):
    x = np.ascontiguousarray(x, np.float64)
    outs = [np.zeros_like(x) for _ in range(5)]
    lib().oracle_det_math(_p(x), x.size, *[_p(o) for o in outs])
    return dict(zip(("sin", "cos", "acos", "exp", "cbrt"), outs))


def pnp_iterative(sc, sub, focal, ppx, ppy, flags_hw, pose6):
    H, W = sc.shape[1], sc.shape[2]
    st = [s // 4 for s in sc.strides]
    fl = np.ascontiguousarray(flags_hw.astype(np.uint8).T)  # x-major [W][H]
    pose = np.array(pose6, np.float64)
    lib().oracle_pnp_iterative(_p(sc), C.c_int64(st[0]), C.c_int64(st[1]), C.c_int64(st[2]), H, W, int(sub), C.c_float(focal),
                               C.c_float(ppx), C.c_float(ppy), _p(fl), _p(pose))
    return pose


def set_options(solver="cholesky", rng="counter", rng_threads=1):
    """solver: 'cholesky' (default, what the GPU kernel runs) | 'svd' (always the eigen pseudo-inverse == cv::solve(DECOMP_SVD));
    rng: 'counter' (default) | 'mt19937' (reference style, thread_rand.cpp: per-thread generators seeded seed + tid once)."""
    lib().oracle_set_options({"cholesky": 0, "svd": 1}[solver], {"counter": 0, "mt19937": 1}[rng], int(rng_threads))


def rng_reset():
    lib().oracle_rng_reset()


def pnp_iterative_pts(obj, img, focal, ppx, ppy, pose6):
    obj = np.ascontiguousarray(obj, np.float32).reshape(-1, 3)
    img = np.ascontiguousarray(img, np.float32).reshape(-1, 2)
    pose = np.array(pose6, np.float64)
    lib().oracle_pnp_iterative_pts(_p(obj), _p(img), obj.shape[0], C.c_float(focal), C.c_float(ppx), C.c_float(ppy), _p(pose))
    return pose

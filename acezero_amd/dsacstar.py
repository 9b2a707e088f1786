"""Drop-in for the reference's `dsacstar` extension module (dsacstar/dsacstar.cpp:898-903).

    import acezero_amd.dsacstar as dsacstar
    inliers = dsacstar.forward_rgb(scene_coordinates_1x3xHxW, out_pose_4x4, hypotheses, threshold, focal, ppX, ppY,
                                   inlier_alpha, max_reproj, subsampling, seed, max_tries)

Same positional arguments, in-place `out_pose` (cam->world) and integer return as register_mapping.py:229-242 uses
them. The work runs on the GPU through libacez.so (acez_register_rgb_host / acez_register_rgb_device); there is no
CPU path. The reference's ThreadRand is seeded once per process and its streams continue across calls
(thread_rand.cpp:13-30); the equivalent here is a per-process call counter used as the frame id of the
counter-based stream, so that successive frames draw different samples but every (seed, call index) is
reproducible. `register_batch` is the batched, device-resident entry the MI355X pipeline should use instead.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N

_ctx = {}
_calls = 0
_verbose = False


def set_verbose(on=True):
    """The reference prints eight progress lines per frame (dsacstar.cpp:101-174); silent by default here, opt-in for parity of the
    console output (stage names as in the reference; the stages run inside one kernel, so there is one time for all of them)."""
    global _verbose
    _verbose = bool(on)
MAX_REF_STEPS = 100  # dsacstar.cpp:47


def _context(max_frames, h, w, device):
    """-> (handle, library that owns it). One cached context per device AND library build: inside N.diag_library() the module-wide
    library is the diagnostics build, and a handle must only ever be driven through the library that created it."""
    L = N.lib()
    key = (device, L._name)
    c = _ctx.get(key)
    if c is None or c["frames"] < max_frames or c["h"] < h or c["w"] < w:
        mf, mh, mw = max(max_frames, c["frames"] if c else 1), max(h, c["h"] if c else 0), max(w, c["w"] if c else 0)
        hnd = C.c_void_p()
        N.check(L.acez_ransac_create(C.byref(hnd), mf, mh, mw, device))   # raises before the old context is touched
        if c is not None:
            del _ctx[key]
            L.acez_ransac_destroy(c["h_"])                                  # waits for the launches still using it
        c = {"h_": hnd, "frames": mf, "h": mh, "w": mw, "lib": L}
        _ctx[key] = c
    return c["h_"], c["lib"]


def _params(hyps, thr, alpha, max_reproj, sub, max_tries):
    return N.RansacParams(int(hyps), int(max_tries), float(thr), float(alpha), float(max_reproj), int(sub), MAX_REF_STEPS, 0)


def reset_call_counter(value=0):
    global _calls
    _calls = int(value)


def forward_rgb(sceneCoordinates, outPose, ransacHypotheses, inlierThreshold, focalLength, ppointX, ppointY, inlierAlpha,
                maxReproj, subSampling, randomSeed, max_hypotheses_tries):
    global _calls
    sc = sceneCoordinates
    if sc.dim() != 4 or sc.shape[0] != 1 or sc.shape[1] != 3 or sc.dtype != torch.float32:
        raise RuntimeError("sceneCoordinates must be a float32 tensor of shape 1x3xHxW")
    if outPose.dim() != 2 or tuple(outPose.shape) != (4, 4) or outPose.dtype != torch.float32:
        raise RuntimeError("outPose must be a float32 tensor of shape 4x4")
    if not torch.cuda.is_available():
        raise RuntimeError("dsacstar.forward_rgb: no GPU visible; the MI355X implementation has no CPU path")
    H, W = int(sc.shape[2]), int(sc.shape[3])
    frame_id = _calls
    _calls += 1
    dev = sc.device.index if sc.is_cuda else torch.cuda.current_device()
    ctx, L = _context(1, H, W, dev)
    prm = _params(ransacHypotheses, inlierThreshold, inlierAlpha, maxReproj, subSampling, max_hypotheses_tries)
    intr = N.Intrinsics(float(focalLength), float(ppointX), float(ppointY))
    import time
    t0 = time.perf_counter()
    if _verbose:
        print("Sampling " + str(int(ransacHypotheses)) + " hypotheses.", flush=True)
    if sc.is_cuda:
        poses, inl, _ = register_batch(sc[0][None], [intr], prm, randomSeed, [frame_id], want_masks=False)
        outPose.copy_(poses[0].to(outPose.device))
        count = int(inl[0].item())
        if _verbose:
            print(f"Calculating scores. / Drawing final hypothesis. / Refining winning pose: done in {(time.perf_counter() - t0) * 1e3:.2f}ms. "
                  f"Inliers: {count}", flush=True)
        return count
    pose = np.zeros(16, np.float32)
    inliers = C.c_int32(0)
    st = sc.stride()
    N.check(L.acez_register_rgb_host(ctx, C.c_void_p(sc.data_ptr()), st[1], st[2], st[3], H, W, C.byref(prm), C.byref(intr),
                                           C.c_uint64(int(randomSeed)), C.c_uint64(frame_id), pose.ctypes.data_as(C.c_void_p),
                                           C.byref(inliers), None))
    outPose.copy_(torch.from_numpy(pose.reshape(4, 4)))
    if _verbose:
        print(f"Calculating scores. / Drawing final hypothesis. / Refining winning pose: done in {(time.perf_counter() - t0) * 1e3:.2f}ms. "
              f"Inliers: {int(inliers.value)}", flush=True)
    return int(inliers.value)


def register_batch(scene_coords, intrinsics, params, seed, frame_ids=None, want_masks=True):
    """scene_coords: CUDA float32 [n,3,H,W]; intrinsics: list of (focal, ppx, ppy) or N.Intrinsics.
    Returns (poses [n,4,4] f32, inliers [n] i32, masks [n,H,W] u8 or None), all CUDA tensors; asynchronous."""
    assert scene_coords.is_cuda and scene_coords.dtype == torch.float32 and scene_coords.dim() == 4 and scene_coords.shape[1] == 3
    sc = scene_coords.contiguous()
    n, _, H, W = sc.shape
    dev = sc.device
    ctx, L = _context(n, H, W, dev.index)
    if not isinstance(params, N.RansacParams):
        params = _params(**params)
    arr = (N.Intrinsics * n)()
    for i, it in enumerate(intrinsics):
        arr[i] = it if isinstance(it, N.Intrinsics) else N.Intrinsics(float(it[0]), float(it[1]), float(it[2]))
    ids = None
    if frame_ids is not None:
        ids = (C.c_uint64 * n)(*[int(x) for x in frame_ids])
    poses = torch.empty(n, 4, 4, dtype=torch.float32, device=dev)
    inl = torch.empty(n, dtype=torch.int32, device=dev)
    masks = torch.empty(n, H, W, dtype=torch.uint8, device=dev) if want_masks else None
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        N.check(L.acez_register_rgb_device(ctx, C.c_void_p(sc.data_ptr()), n, H, W, C.byref(params), arr, C.c_uint64(int(seed)),
                                                 ids, C.c_void_p(poses.data_ptr()), C.c_void_p(inl.data_ptr()),
                                                 C.c_void_p(masks.data_ptr()) if masks is not None else None, stream))
    return poses, inl, masks


def debug_fetch(n, hyps, device=None):
    dev = torch.cuda.current_device() if device is None else device
    L = N.lib()
    ctx = _ctx[(dev, L._name)]["h_"]
    hp = np.zeros((n, hyps, 6)); sc = np.zeros((n, hyps)); best = np.zeros(n, np.int32); ref = np.zeros((n, 6))
    N.check(L.acez_ransac_debug_fetch(ctx, n, hyps, hp.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p),
                                            best.ctypes.data_as(C.c_void_p), ref.ctypes.data_as(C.c_void_p)))
    return {"hyp_poses": hp, "scores": sc, "best": best, "refined": ref}

"""ctypes binding of libacez.so (include/acez.h).  No compute happens in Python: every call below enqueues HIP
kernels through the C ABI.  There is deliberately NO fallback: if the library is missing it is built with hipcc,
and if that fails the import error propagates."""
import contextlib
import ctypes as C
import os

# torch FIRST: its wheel bundles its own libamdhip64 / libhsa-runtime64. Loaded first, libacez.so's HIP dependency resolves to
# that same runtime (one HSA instance per process, shared streams and allocations). Loaded second, the process would hold two
# HIP runtimes and the one initialised last sees no device.
import torch  # noqa: F401

from . import build as _build

_lib = None

ACEZ_OK = 0
ERRORS = {-1: "ACEZ_ERR_INVALID", -2: "ACEZ_ERR_HIP", -3: "ACEZ_ERR_NODEVICE", -4: "ACEZ_ERR_NAN"}


class RansacParams(C.Structure):
    _fields_ = [("hypotheses", C.c_int32), ("max_tries", C.c_int32), ("inlier_threshold", C.c_float),
                ("inlier_alpha", C.c_float), ("max_reproj", C.c_float), ("subsampling", C.c_int32),
                ("max_ref_steps", C.c_int32), ("reserved", C.c_int32)]


class Intrinsics(C.Structure):
    _fields_ = [("focal", C.c_float), ("ppx", C.c_float), ("ppy", C.c_float)]


class HeadDesc(C.Structure):
    _fields_ = [("num_head_blocks", C.c_int32), ("use_homogeneous", C.c_int32), ("mean", C.c_float * 3),
                ("max_inv_scale", C.c_float), ("min_inv_scale", C.c_float), ("h_beta", C.c_float)]


class TrainConfig(C.Structure):
    _fields_ = [("head", HeadDesc), ("max_batch", C.c_int32), ("global_batch", C.c_int32), ("loss_type", C.c_int32),
                ("soft_clamp", C.c_float), ("soft_clamp_min", C.c_float), ("circle_schedule", C.c_int32),
                ("hard_clamp", C.c_float), ("depth_min", C.c_float), ("depth_max", C.c_float), ("depth_target", C.c_float),
                ("inlier_px_threshold", C.c_float), ("schedule", C.c_int32), ("iterations", C.c_int32),
                ("lr_min", C.c_double), ("lr_max", C.c_double), ("warmup_iterations", C.c_int32), ("warmup_lr", C.c_double),
                ("cooldown_iterations", C.c_int32), ("cooldown_trigger_percent", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double), ("refine_calibration", C.c_int32),
                ("focal_init", C.c_float), ("calib_lr", C.c_double), ("pose_refinement", C.c_int32), ("pose_refinement_wait", C.c_int32),
                ("pose_refinement_lr", C.c_double), ("pose_refinement_weight", C.c_float), ("pose_refinement_ortho", C.c_int32),
                ("compute_dtype", C.c_int32), ("inference_only", C.c_int32)]


class ParamBuffers(C.Structure):
    _fields_ = [("d_params", C.c_void_p), ("d_adam_m", C.c_void_p), ("d_adam_v", C.c_void_p), ("d_grad", C.c_void_p),
                ("n_params", C.c_int64), ("d_pose_params", C.c_void_p), ("d_pose_m", C.c_void_p), ("d_pose_v", C.c_void_p),
                ("n_pose_params", C.c_int64)]


class TrainBuffer(C.Structure):
    _fields_ = [("d_features", C.c_void_p), ("d_target_px", C.c_void_p), ("d_view_idx", C.c_void_p), ("n_patches", C.c_int64),
                ("d_view_aug_inv", C.c_void_p), ("d_view_K", C.c_void_p), ("d_view_Kinv", C.c_void_p),
                ("d_view_image", C.c_void_p), ("n_views", C.c_int32), ("d_image_pose_inv", C.c_void_p), ("n_images", C.c_int32),
                ("d_target_crds", C.c_void_p)]


class TrainState(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("max_iterations", C.c_int32), ("in_cooldown", C.c_int32), ("nan_flag", C.c_int32),
                ("lr", C.c_double), ("last_loss", C.c_float), ("last_batch_inliers", C.c_float), ("focal_scale", C.c_double),
                ("grad_scale", C.c_float), ("opt_steps", C.c_int32)]


# every symbol include/acez.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "acez_last_error": (C.c_char_p, []),
    "acez_version": (C.c_char_p, []),
    "acez_device_count": (C.c_int, []),
    "acez_ransac_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
    "acez_ransac_destroy": (None, [C.c_void_p]),
    "acez_register_rgb_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(RansacParams),
                                           C.POINTER(Intrinsics), C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "acez_register_rgb_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                         C.POINTER(RansacParams), C.POINTER(Intrinsics), C.c_uint64, C.c_uint64, C.c_void_p,
                                         C.POINTER(C.c_int32), C.c_void_p]),
    "acez_ransac_debug_fetch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "acez_head_num_params": (C.c_int64, [C.POINTER(HeadDesc)]),
    "acez_trainer_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(TrainConfig), C.POINTER(ParamBuffers), C.c_int]),
    "acez_trainer_destroy": (None, [C.c_void_p]),
    "acez_trainer_set_buffer": (C.c_int, [C.c_void_p, C.POINTER(TrainBuffer)]),
    "acez_trainer_sync_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "acez_train_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "acez_train_update": (C.c_int, [C.c_void_p, C.c_void_p]),
    "acez_train_update_next": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "acez_train_update_layers": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "acez_trainer_export_weights16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "acez_trainer_import_weights16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "acez_trainer_import_weights16_all": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "acez_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "acez_train_step_next": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "acez_trainer_get_state": (C.c_int, [C.c_void_p, C.POINTER(TrainState), C.c_void_p]),
    "acez_trainer_get_log": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "acez_trainer_last_scene_coords": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "acez_trainer_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "acez_trainer_get_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "acez_trainer_seq_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "acez_trainer_debug_read": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "acez_trainer_get_poses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "acez_head_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "acez_head_forward_maps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "acez_buffer_warp_views": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "acez_buffer_sample_views": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                           C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "acez_encoder_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int]),
    "acez_encoder_destroy": (None, [C.c_void_p]),
    "acez_encoder_output_size": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "acez_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "acez_point_cloud_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                          C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
}


def _load(path, lenient=False):
    if not os.path.exists(path):
        raise RuntimeError("libacez.so is missing and could not be built: the HIP extension is mandatory")
    L = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        if lenient and not hasattr(L, name):
            continue
        fn = getattr(L, name)  # AttributeError here means the library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    return L


def lib():
    """Load (building first if needed) libacez.so and declare every prototype."""
    global _lib
    if _lib is None:
        other = os.environ.get("ACEZ_LIB")   # diagnostics only (tools/lib_variant.sh): time another build of the library on the same box
        if other == "diag":
            other = _build.build(diag=True)
        _lib = _load(other or _build.build(), lenient=bool(other))
    return _lib


_diag = None


@contextlib.contextmanager
def diag_library():
    """tests/ and tools/ only: inside the block lib() is the diagnostics build (libacez_diag.so = the same sources with -DACEZ_DIAG), the
    only build in which the ACEZ_* ablation switches, the measured-and-rejected kernels and the fault-injection hooks exist. Every
    handle-owning object pins the library that created its handle (HeadTrainer.lib, Encoder.lib, the RANSAC contexts of dsacstar.py are
    cached per (device, library)), so an object created under one build is never driven through the other, inside or outside the block."""
    global _lib, _diag
    if _diag is None:
        _diag = _load(_build.build(diag=True))
    saved, _lib = _lib, _diag
    try:
        yield _diag
    finally:
        _lib = saved


class AcezError(RuntimeError):
    pass


def check(rc):
    if rc != ACEZ_OK:
        msg = lib().acez_last_error().decode("utf-8", "replace")
        raise AcezError("%s: %s" % (ERRORS.get(rc, rc), msg))

"""Host side of the head path: device buffers (torch tensors as plain HBM allocations) + the C-ABI trainer.

Mirrors the reference's seams:
  HeadTrainer.state_dict()/load_state_dict()  ->  Head.state_dict() keys of ace_network.py (ace_trainer.py:690-693)
  HeadTrainer.step(indices)                   ->  TrainerACE.training_step (ace_trainer.py:499-679)
  HeadTrainer.get_scene_coordinates(features) ->  Regressor.get_scene_coordinates (ace_network.py:262-263)
"""
import ctypes as C
import math

import numpy as np
import torch

from . import DEFAULT_DTYPE  # when neither an argument nor $ACEZ_DTYPE names a format
from . import _native as N

LOSS_TYPES = {"tanh": 0, "dyntanh": 1, "l1": 2, "l1+sqrt": 3, "l1+logl1": 4, "l1+log": 4}
SCHEDULES = {"constant": 0, "1cyclepoly": 1, "circle": 2}
DTYPES = {"bf16": 0, "fp16": 1}   # acez_train_config.compute_dtype / acez_encoder_create's compute_dtype


def resolve_dtype(dtype=None):
    """The 16-bit operand format of a context: an explicit "bf16" / "fp16", else $ACEZ_DTYPE, else DEFAULT_DTYPE. "fp32" (train_ace.py
    --use_half False) is rejected: it is not implemented and must not silently run in another precision."""
    import os
    dtype = (dtype or os.environ.get("ACEZ_DTYPE") or DEFAULT_DTYPE).lower()
    if dtype in ("fp32", "float32"):
        raise NotImplementedError("fp32 arithmetic (--use_half False) is not implemented; choose dtype='fp16' (the reference's "
                                  "autocast format) or 'bf16'")
    if dtype not in DTYPES:
        raise ValueError("dtype must be 'bf16' or 'fp16'")
    return dtype


def torch_dtype(dtype):
    return torch.float16 if dtype == "fp16" else torch.bfloat16


def layer_names(num_head_blocks):
    names = ["res3_conv1", "res3_conv2", "res3_conv3"]
    for b in range(num_head_blocks):
        names += [f"{b}c0", f"{b}c1", f"{b}c2"]
    return names + ["fc1", "fc2"]


POSE_MLP_PARAMS = 70924
POSE_LAYERS = [("head_skip", 128, 12), ("conv1", 128, 12), ("conv2", 128, 128), ("conv3", 128, 128), ("fc1", 128, 128),
               ("fc2", 128, 128), ("fc3", 12, 128)]   # PoseNetwork(0, 128).named_parameters() order (refine_poses.py:21-51)


def init_pose_network(seed):
    """nn.Conv2d default initialisation (kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and
    bias) of PoseNetwork(0, 128), flat in named_parameters() order."""
    g = torch.Generator().manual_seed(int(seed))
    parts = []
    for _, o, k in POSE_LAYERS:
        bound = 1.0 / math.sqrt(k)
        parts.append((torch.rand(o * k, generator=g) * 2 - 1) * bound)
        parts.append((torch.rand(o, generator=g) * 2 - 1) * bound)
    return torch.cat(parts).float()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def epoch_permutations(n, seed, device):
    """Endless stream of per-epoch permutations of the buffer rows (TrainerACE.run_epoch, ace_trainer.py:466) as int64 device
    tensors. The reference draws them with torch.randperm on the CPU; measured here that costs 100 us per training step
    (12 ms per 600 k rows, 55 % on top of an 8 M-row epoch) because the host cannot run ahead of the GPU across the draw, so
    the draw is a device-side torch.randperm from a device generator seeded like the reference's (seed = base_seed + 8191):
    same law (a uniform permutation per epoch), different stream."""
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    # Small buffers (a seed round of ace_zero.py maps ONE image: 10 240 rows = two steps per epoch): one randperm call per epoch is several
    # small launches per two training steps, so the permutations of `group` epochs come out of one draw -- the ranks of independent uniform
    # doubles (53 random bits: no ties to break), row by row. Same law, one batched sort instead of `group` small ones.
    group = max(1, 262144 // max(n, 1))
    while True:
        if group == 1:
            yield torch.randperm(n, generator=gen, device=device)
        else:
            block = torch.rand((group, n), generator=gen, device=device, dtype=torch.float64).argsort(dim=1)
            for k in range(group):
                yield block[k]


def epoch_batches(n, batch, seed, device):
    """The walk of TrainerACE.run_epoch (ace_trainer.py:454-497) as an endless stream of (rows of this step, rows of the NEXT step): per
    epoch one permutation of the buffer rows in slices of `batch` (a tail that does not fill a batch is dropped). The announcement crosses
    the epoch boundary -- the next epoch's permutation is drawn an epoch ahead -- so every step of the loop is an
    acez_train_step_next whose successor's gather rides in its optimiser launch (a seed round has two steps per epoch: every other step
    used to be unannounced). The second tensor of a pair has the data pointer of the first tensor of the following pair."""
    perms = epoch_permutations(n, seed, device)
    nb = n // batch
    if nb < 1:
        raise ValueError(f"training buffer of {n} rows is smaller than one batch ({batch})")
    cur = next(perms)
    while True:
        nxt = next(perms)
        for b in range(nb):
            yield cur[b * batch:(b + 1) * batch], (cur[(b + 1) * batch:(b + 2) * batch] if b + 1 < nb else nxt[:batch])
        cur = nxt


class HeadTrainer:
    """One head + its optimiser/schedule state on one GPU, driven through libacez.so."""

    def __init__(self, mean, *, num_head_blocks=1, use_homogeneous=True, max_batch=5120, global_batch=None,
                 loss_type="dyntanh", soft_clamp=50.0, soft_clamp_min=1.0, circle_schedule=True, hard_clamp=1000.0,
                 depth_min=0.1, depth_max=1000.0, depth_target=10.0, inlier_px_threshold=10.0, schedule="circle",
                 iterations=25000, lr_min=0.0005, lr_max=0.005, warmup_iterations=1000, warmup_lr=0.0005,
                 cooldown_iterations=5000, cooldown_trigger_percent=0.7, refine_calibration=False, focal_init=0.0,
                 calib_lr=0.001, pose_refinement="none", pose_refinement_wait=0, pose_refinement_lr=0.001,
                 pose_refinement_weight=0.1, refinement_ortho="gram-schmidt", pose_seed=0, initial_poses=None, homogeneous_min_scale=0.01, homogeneous_max_scale=4.0,
                 device=None, dtype=None, inference_only=False):
        """dtype: 16-bit operand format of the head's GEMMs, "bf16" (default) or "fp16" (the reference's autocast format,
        ace_trainer.py:517-518); None reads ACEZ_DTYPE. "fp32" (train_ace.py --use_half False) is rejected: it is not implemented
        and must not silently run in another precision.
        inference_only: a context for get_scene_coordinates / Regressor only -- no gradient or partial-sum buffers on the device, the
        training calls raise (acez_train_config.inference_only)."""
        if not torch.cuda.is_available():
            raise RuntimeError("HeadTrainer needs a GPU: the head kernels are HIP only (no CPU fallback)")
        self.lib = N.lib()
        self.dtype = dtype = resolve_dtype(dtype)
        self.feature_dtype = torch_dtype(dtype)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.nb, self.homog = int(num_head_blocks), bool(use_homogeneous)
        self.L = 3 + 3 * self.nb + 2
        self.no = 4 if self.homog else 3
        self.mean = torch.as_tensor(mean, dtype=torch.float32).view(3).clone()
        max_scale = torch.tensor([homogeneous_max_scale])
        min_scale = torch.tensor([homogeneous_min_scale])
        self.buffers = {  # ace_network.py:109-118
            "max_scale": max_scale, "min_scale": min_scale, "max_inv_scale": 1.0 / max_scale,
            "h_beta": math.log(2) / (1.0 - 1.0 / max_scale), "min_inv_scale": 1.0 / min_scale,
            "mean": self.mean.view(1, 3, 1, 1).clone(),
        }
        hd = N.HeadDesc(self.nb, int(self.homog), (C.c_float * 3)(*[float(x) for x in self.mean]),
                        float(self.buffers["max_inv_scale"]), float(self.buffers["min_inv_scale"]), float(self.buffers["h_beta"]))
        self.n_params = int(self.lib.acez_head_num_params(C.byref(hd)))
        dev = self.device
        # fp32 masters and AdamW moments: three views of ONE allocation (like the library's arenas, acez_trainer_create: one contiguous
        # virtual range instead of three; each view starts on a 4 KiB boundary)
        npad = (self.n_params + 1023) // 1024 * 1024
        self._state = torch.zeros(3 * npad, dtype=torch.float32, device=dev)
        self.params, self.adam_m, self.adam_v = (self._state[i * npad:i * npad + self.n_params] for i in range(3))
        if pose_refinement not in ("none", "naive", "mlp"):
            raise ValueError("pose_refinement must be 'none', 'naive' or 'mlp'")
        self.pose_mlp = pose_refinement == "mlp"
        self.pose_naive = pose_refinement == "naive"
        if self.pose_naive and initial_poses is None:
            raise ValueError("pose_refinement='naive' needs initial_poses [n_images,3,4] (world->cam), the learnable parameters")
        self.n_pose = POSE_MLP_PARAMS if self.pose_mlp else (12 * len(initial_poses) if self.pose_naive else 0)
        # gradient bucket: head gradients, {loss, inliers, dfocal, pad}, then the pose-network gradients
        self.grad = torch.zeros(self.n_params + 4 + self.n_pose, dtype=torch.float32, device=dev)
        self.pose_params = self.pose_m = self.pose_v = None
        if self.pose_mlp or self.pose_naive:
            self.pose_params = (init_pose_network(pose_seed) if self.pose_mlp else
                                torch.as_tensor(initial_poses, dtype=torch.float32).reshape(-1).clone()).to(dev)
            self.pose_m = torch.zeros_like(self.pose_params)
            self.pose_v = torch.zeros_like(self.pose_params)
        self.max_batch = int(max_batch)
        self.global_batch = int(global_batch or max_batch)
        cfg = N.TrainConfig()
        cfg.head = hd
        cfg.max_batch, cfg.global_batch = self.max_batch, self.global_batch
        cfg.loss_type = LOSS_TYPES[loss_type]
        cfg.soft_clamp, cfg.soft_clamp_min, cfg.circle_schedule = soft_clamp, soft_clamp_min, int(circle_schedule)
        cfg.hard_clamp, cfg.depth_min, cfg.depth_max, cfg.depth_target = hard_clamp, depth_min, depth_max, depth_target
        cfg.inlier_px_threshold = inlier_px_threshold
        cfg.schedule, cfg.iterations = SCHEDULES[schedule], int(iterations)
        cfg.lr_min, cfg.lr_max, cfg.warmup_iterations, cfg.warmup_lr = lr_min, lr_max, int(warmup_iterations), warmup_lr
        cfg.cooldown_iterations, cfg.cooldown_trigger_percent = int(cooldown_iterations), cooldown_trigger_percent
        cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay = 0.9, 0.999, 1e-8, 1e-2  # torch.optim.AdamW defaults
        cfg.refine_calibration, cfg.focal_init, cfg.calib_lr = int(refine_calibration), float(focal_init), calib_lr
        cfg.pose_refinement = 2 if self.pose_mlp else (1 if self.pose_naive else 0)
        cfg.pose_refinement_wait, cfg.pose_refinement_lr = int(pose_refinement_wait), float(pose_refinement_lr)
        cfg.pose_refinement_weight = float(pose_refinement_weight)
        if refinement_ortho not in ("gram-schmidt", "procrustes"):
            raise ValueError("refinement_ortho must be 'gram-schmidt' or 'procrustes'")
        cfg.pose_refinement_ortho = 1 if refinement_ortho == "procrustes" else 0
        cfg.compute_dtype = DTYPES[self.dtype]
        cfg.inference_only = 1 if inference_only else 0
        pb = N.ParamBuffers(_ptr(self.params), _ptr(self.adam_m), _ptr(self.adam_v), _ptr(self.grad), self.n_params,
                            _ptr(self.pose_params), _ptr(self.pose_m), _ptr(self.pose_v), self.n_pose)
        h = C.c_void_p()
        N.check(self.lib.acez_trainer_create(C.byref(h), C.byref(cfg), C.byref(pb), self.device.index))
        self._h = h
        self._buf = None
        self.iterations = int(iterations)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.acez_trainer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- parameters <-> Head.state_dict()
    def _views(self):
        out, o = {}, 0
        for name in layer_names(self.nb):
            out[name + ".weight"] = self.params[o:o + 262144].view(512, 512, 1, 1)
            o += 262144
            out[name + ".bias"] = self.params[o:o + 512]
            o += 512
        out["fc3.weight"] = self.params[o:o + self.no * 512].view(self.no, 512, 1, 1)
        o += self.no * 512
        out["fc3.bias"] = self.params[o:o + self.no]
        return out

    def _require_whole_masters(self, what):
        # Under parallel.ShardedDataParallel a rank keeps only ITS layers' fp32 masters / AdamW moments current (update_layers with a
        # partial range); the others go stale until gather_masters(). Reading or recasting them in that state would silently use stale
        # weights, so it is an error.
        if getattr(self, "_masters_partial", False):
            raise RuntimeError(f"{what}: the fp32 masters of layers this rank does not own are stale (sharded data-parallel update); "
                               "call ShardedDataParallel.gather_masters() first")

    def masters_synced(self):
        """Called by whoever has made every layer's masters current again (ShardedDataParallel.gather_masters)."""
        self._masters_partial = False

    def state_dict(self):
        """Same keys / shapes as ace_network.Head.state_dict() (fp32; the trainer saves .half(), ace_trainer.py:690)."""
        self._require_whole_masters("state_dict()")
        sd = {k: v.clone() for k, v in self.buffers.items() if self.homog or k == "mean"}
        sd.update({k: v.detach().clone() for k, v in self._views().items()})
        return sd

    def load_state_dict(self, sd):
        views = self._views()
        for k, v in views.items():
            v.copy_(sd[k].to(torch.float32).view_as(v))
        if "mean" in sd and not torch.allclose(sd["mean"].float().view(3).cpu(), self.mean):
            raise ValueError("mean of the checkpoint differs from the mean this trainer was created with")
        self._masters_partial = False      # every master has just been overwritten
        self.sync_weights()

    def load_flat(self, flat):
        self.params.copy_(torch.as_tensor(flat, dtype=torch.float32).to(self.device))
        self._masters_partial = False
        self.sync_weights()

    def sync_weights(self):
        """Recast the 16-bit compute copies from the fp32 masters (after the caller rewrote them)."""
        self._require_whole_masters("sync_weights()")   # would clobber the imported 16-bit copies of foreign layers with stale masters
        N.check(self.lib.acez_trainer_sync_weights(self._h, _stream()))

    # ---------------------------------------------------------------- training buffer
    def set_buffer(self, features, target_px, view_idx, view_aug_inv, view_K, view_Kinv, view_image, image_pose_inv, target_crds=None):
        """All arguments are arrays/tensors in the de-duplicated layout of acez_train_buffer (include/acez.h)."""
        dev = self.device
        t = lambda x, dt: torch.as_tensor(x).to(device=dev, dtype=dt).contiguous()
        self._buf = {
            "features": t(features, self.feature_dtype), "target_px": t(target_px, torch.float32), "view_idx": t(view_idx, torch.int32),
            "view_aug_inv": t(view_aug_inv, torch.float32), "view_K": t(view_K, torch.float32),
            "view_Kinv": t(view_Kinv, torch.float32), "view_image": t(view_image, torch.int32),
            "image_pose_inv": t(image_pose_inv, torch.float32),
            "target_crds": t(target_crds, torch.float32) if target_crds is not None else None,   # use_depth mode
        }
        b = self._buf
        assert b["features"].shape[1] == 512
        tb = N.TrainBuffer(_ptr(b["features"]), _ptr(b["target_px"]), _ptr(b["view_idx"]), b["features"].shape[0],
                           _ptr(b["view_aug_inv"]), _ptr(b["view_K"]), _ptr(b["view_Kinv"]), _ptr(b["view_image"]),
                           b["view_aug_inv"].shape[0], _ptr(b["image_pose_inv"]), b["image_pose_inv"].shape[0], _ptr(b["target_crds"]))
        N.check(self.lib.acez_trainer_set_buffer(self._h, C.byref(tb)))

    @property
    def buffer_size(self):
        return 0 if self._buf is None else int(self._buf["features"].shape[0])

    # ---------------------------------------------------------------- the step
    def backward(self, indices):
        """indices: int64 CUDA tensor of buffer rows (<= max_batch). Asynchronous."""
        assert indices.dtype == torch.int64 and indices.is_cuda and indices.is_contiguous()
        N.check(self.lib.acez_train_backward(self._h, _ptr(indices), int(indices.numel()), _stream()))

    def update(self, next_indices=None):
        """AdamW + schedule. next_indices: the rows of the NEXT backward() call, if known (data-parallel ranks draw the same epoch
        permutation): gathered inside the optimiser's launch; pass the very same tensor to that backward()."""
        if next_indices is None or next_indices.numel() == 0:
            N.check(self.lib.acez_train_update(self._h, _stream()))
        else:
            assert next_indices.dtype == torch.int64 and next_indices.is_cuda and next_indices.is_contiguous()
            self._next_keepalive = next_indices     # the device pointer must stay valid until the next call has consumed it
            N.check(self.lib.acez_train_update_next(self._h, _ptr(next_indices), int(next_indices.numel()), _stream()))

    # ---- sharded data-parallel update (parallel.ShardedDataParallel)
    LAYER_STRIDE = 262144 + 512          # floats of one wide layer (weight + bias) in the flat parameter / gradient vectors

    def update_layers(self, layer_lo, layer_hi):
        """AdamW on the weight matrices of wide layers [layer_lo, layer_hi) + all small parameters + the schedule bookkeeping."""
        N.check(self.lib.acez_train_update_layers(self._h, int(layer_lo), int(layer_hi), _stream()))
        if (int(layer_lo), int(layer_hi)) != (0, self.L):
            self._masters_partial = True

    def new_weights16_buffer(self):
        """[L, 512 * 512 / 2] device tensor of int32 words (two 16-bit weights each; a dtype every torch.distributed backend moves): the
        all-gather buffer of the compute copies."""
        return torch.empty((self.L, 131072), dtype=torch.int32, device=self.device)

    def export_weights16(self, layer_lo, layer_hi, dst):
        assert dst.is_cuda and dst.is_contiguous() and dst.numel() * dst.element_size() == (layer_hi - layer_lo) * 524288
        if layer_hi > layer_lo:
            N.check(self.lib.acez_trainer_export_weights16(self._h, int(layer_lo), int(layer_hi), _ptr(dst), _stream()))

    def import_weights16(self, layer_lo, layer_hi, src):
        assert src.is_cuda and src.is_contiguous() and src.numel() * src.element_size() == (layer_hi - layer_lo) * 524288
        if layer_hi > layer_lo:
            N.check(self.lib.acez_trainer_import_weights16(self._h, int(layer_lo), int(layer_hi), _ptr(src), _stream()))

    def import_weights16_all(self, own_lo, own_hi, src_all):
        """One launch for the receiving side of the all-gather: every layer outside [own_lo, own_hi) from `src_all` ([L, 131072] int32)."""
        assert src_all.is_cuda and src_all.is_contiguous() and src_all.numel() * src_all.element_size() == self.L * 524288
        N.check(self.lib.acez_trainer_import_weights16_all(self._h, int(own_lo), int(own_hi), _ptr(src_all), _stream()))

    def master_tensors(self):
        """The flat fp32 vectors whose wide-layer ranges an owner rank keeps current under the sharded update."""
        return [self.params, self.adam_m, self.adam_v]

    def step(self, indices, next_indices=None):
        """One fused training step. next_indices: the batch of the FOLLOWING step() call, if known (run_epoch walks one permutation):
        it is gathered inside this step's optimiser launch; pass the very same tensor to the next call."""
        assert indices.dtype == torch.int64 and indices.is_cuda and indices.is_contiguous()
        if next_indices is None:
            N.check(self.lib.acez_train_step(self._h, _ptr(indices), int(indices.numel()), _stream()))
        else:
            assert next_indices.dtype == torch.int64 and next_indices.is_cuda and next_indices.is_contiguous()
            self._next_keepalive = next_indices     # the device pointer must stay valid until the next call has consumed it
            N.check(self.lib.acez_train_step_next(self._h, _ptr(indices), int(indices.numel()), _ptr(next_indices), int(next_indices.numel()),
                                                  _stream()))

    def state(self):
        st = N.TrainState()
        rc = self.lib.acez_trainer_get_state(self._h, C.byref(st), _stream())
        if rc not in (0, -4):
            N.check(rc)
        return {"iteration": st.iteration, "max_iterations": st.max_iterations, "in_cooldown": bool(st.in_cooldown),
                "nan": bool(st.nan_flag), "lr": st.lr, "loss": st.last_loss, "batch_inliers": st.last_batch_inliers,
                "focal_scale": st.focal_scale, "grad_scale": st.grad_scale, "opt_steps": st.opt_steps}

    def seq_status(self):
        """One-launch GEMM chains (rowseq_kernel): {'enabled', 'probe' (placement probe at creation: 1 passed, 0 failed, -1 not run),
        'faults' (fall-backs to per-layer launches taken after an expired hand-off poll; state() performs them)}."""
        en, pr, fl = C.c_int(0), C.c_int(0), C.c_int(0)
        N.check(self.lib.acez_trainer_seq_status(self._h, C.byref(en), C.byref(pr), C.byref(fl)))
        return {"enabled": bool(en.value), "probe": pr.value, "faults": fl.value}

    def log(self, first, count):
        loss = np.zeros(count, np.float32)
        inl = np.zeros(count, np.float32)
        N.check(self.lib.acez_trainer_get_log(self._h, first, count, loss.ctypes.data_as(C.c_void_p),
                                              inl.ctypes.data_as(C.c_void_p), _stream()))
        return loss, inl

    def last_scene_coords(self, n):
        out = np.zeros((n, 3), np.float32)
        N.check(self.lib.acez_trainer_last_scene_coords(self._h, out.ctypes.data_as(C.c_void_p), n, _stream()))
        return out

    KERNEL_CLASSES = ("sched", "gather", "gemm_fwd", "loss", "gemm_dgrad", "wgrad", "grad_reduce", "adamw")

    def set_profiling(self, on):
        N.check(self.lib.acez_trainer_set_profiling(self._h, int(bool(on))))

    def get_profile(self):
        """{class: (total_ms, launches)} measured with HIP events on the launch stream since profiling was enabled."""
        ms = np.zeros(8, np.float32)
        cnt = np.zeros(8, np.int32)
        N.check(self.lib.acez_trainer_get_profile(self._h, ms.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.KERNEL_CLASSES)}

    DEBUG_KINDS = {"out": 0, "dZ": 1, "R": 2, "slab": 3, "bias_partials": 4, "mask": 9}

    def debug_read(self, kind, index, rows):
        """Intermediate buffer of the last backward call (tests): 'out' / 'dZ' / 'R' -> uint16 [rows,512] (bf16 bit patterns),
        'slab' -> float32 [n_wide], 'bias_partials' -> float32 [rows,512], 'mask' -> uint32 [80-row tiles, 2048] (the lane-private ReLU
        mask bits a training forward leaves per layer; the pre-residual activation of a block's last layer is not kept in training)."""
        if kind in ("out", "dZ", "R"):
            out = np.zeros((rows, 512), np.uint16)
        elif kind == "mask":
            out = np.zeros(((rows + 79) // 80, 2048), np.uint32)
        elif kind == "slab":
            out = np.zeros(self.L * 262656, np.float32)
        else:
            out = np.zeros((rows, 512), np.float32)
        N.check(self.lib.acez_trainer_debug_read(self._h, self.DEBUG_KINDS[kind], int(index), out.ctypes.data_as(C.c_void_p),
                                                 out.nbytes, _stream()))
        return out

    def current_poses(self):
        """Refined world->cam poses [n_images,3,4] (PoseRefiner.get_all_current_poses, refine_poses.py:184-210)."""
        n = int(self._buf["image_pose_inv"].shape[0])
        out = np.zeros((n, 3, 4), np.float32)
        N.check(self.lib.acez_trainer_get_poses(self._h, out.ctypes.data_as(C.c_void_p), _stream()))
        return out

    # ---------------------------------------------------------------- inference
    def get_scene_coordinates(self, features):
        """features: [n,512] CUDA tensor (any float dtype) -> [n,3] float32 CUDA tensor."""
        f = features.to(device=self.device, dtype=self.feature_dtype).contiguous()
        out = torch.empty(f.shape[0], 3, dtype=torch.float32, device=self.device)
        N.check(self.lib.acez_head_forward(self._h, _ptr(f), int(f.shape[0]), _ptr(out), _stream()))
        return out

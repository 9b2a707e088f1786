"""One persistent process per GPU for the whole ACE0 reconstruction loop.

ace_zero.py drives the reconstruction by spawning `train_ace.py` / `register_mapping.py` once per iteration
(ace_zero_util.py:11-52, ace_zero.py:195-196,235,274,307): every call pays interpreter + torch import, context creation,
data-loader start-up, and re-encodes every image.  Here the same loop (seed trials -> best seed -> map / register rounds with
warm start, pose-MLP and focal refinement -> stopping criteria -> final refit) runs in-process on frames that are already in
memory: HIP context, encoder weights and the ENCODER FEATURES of all frames (4.9 MB per 480x640 frame in 16 bits: 1000 frames
= 4.9 GB of the 288 GB) stay resident, so a mapping round is buffer sampling + the training kernels and a registration
round is head + RANSAC on cached features.

    ses = ReconstructionSession(encoder_state_dict, images_n1hw, depth=seed_depth_nhw)     # opt = default_options(...)
    result = ses.reconstruct()          # {"poses": cam->world [n,4,4], "confidence": [n], "focal": f, "head": state_dict, ...}

Option names and defaults are ace_zero.py's (:41-177).  What is NOT here is file decoding / resizing (dataset.py, done by the
caller: frames arrive as normalised grey tensors) and ZoeDepth: a seed needs a depth map for its image (ace_zero.py's
--depth_files path; the reference falls back to a network download otherwise).  With use_aug (the default, as in the reference)
every buffer pass re-encodes freshly rotated / rescaled / brightness-jittered views (dataset.py:268-356) warped on the GPU.
With torch.distributed initialised (torchrun ace_zero.py) the frames, the buffer and the registration are sharded over the ranks
and mapping is data parallel (parallel.py).  Frames must fit the DSAC* kernel: at most 16384 scene coordinates (H/8 x W/8).
There is no CPU fallback.
"""
import logging
import math
import os
import time
from types import SimpleNamespace

import numpy as np
import torch

from . import _native as N
from . import dsacstar
from .encoder import Encoder, output_size
import torch.distributed as dist

from .head import HeadTrainer, _ptr, _stream, epoch_batches, epoch_permutations
from .parallel import epoch_local_batches, gather_registrations, make_data_parallel, rank_world

_logger = logging.getLogger("acezero_amd.session")


MAX_SCENE_COORDINATES = 16384   # per frame: the DSAC* kernel's inlier word (ransac_api.hip MAX_ROWS x 256 threads)


def check_frame_size(h, w):
    """The registration kernel handles frames of up to 16384 scene coordinates (480 x 2184, 768 x 1360, ...); the reference's CPU loop
    has no limit. Raise early, with the remedy, instead of failing after every frame has been encoded."""
    oh, ow = output_size(h, w)
    if oh * ow > MAX_SCENE_COORDINATES:
        raise RuntimeError(f"{h} x {w} frames give {oh} x {ow} = {oh * ow} scene coordinates, the DSAC* kernel registers at most "
                           f"{MAX_SCENE_COORDINATES} per frame: lower --image_resolution (dataset.py:40 rescales the short side to it)")


def default_options(**over):
    """ace_zero.py:41-177 (the flags that reach the hot path) + the train_ace.py / register_mapping.py defaults they rely on."""
    o = dict(iterations_max=100, registration_threshold=0.99, relative_registration_threshold=0.01, final_refine=True, final_refit=True,
             final_refit_posewait=5000, refit_iterations=25000, registration_confidence=500, try_seeds=5, seed_iterations=10000,
             warmstart=True, seed_network=None, export_point_cloud=False, dense_point_cloud=False, refinement="mlp", refinement_ortho="gram-schmidt", pose_refinement_wait=0, pose_refinement_lr=0.001,
             refine_calibration=True, use_external_focal_length=-1.0, learning_rate_schedule="1cyclepoly", learning_rate_max=0.003,
             cooldown_iterations=5000, cooldown_threshold=0.7, num_head_blocks=1, max_dataset_passes=10, repro_loss_type="tanh",
             repro_loss_hard_clamp=1000, repro_loss_soft_clamp=50, ransac_iterations=32, ransac_threshold=10.0, random_seed=1305,
             # train_ace.py defaults ace_zero.py does not override
             iterations=25000, learning_rate_min=0.0005, learning_rate_warmup_iterations=1000, learning_rate_warmup_learning_rate=0.0005,
             max_training_buffer_size=8000000, samples_per_image=1024, batch_size=5120, base_seed=2089, register_seed=1305,
             max_estimates_seed_scoring=1000,
             # train_ace.py augmentation flags (:174-182) + dataset.py:38-41; applied on the device to the resident frames
             use_aug=True, aug_rotation=15, aug_scale=1.5, aug_black_white=0.1,
             # train_ace.py flags only its own command line changes
             use_homogeneous=True, repro_loss_soft_clamp_min=1, repro_loss_schedule="circle", refine_calibration_lr=0.001,
             pose_refinement_weight=0.1, learning_rate_cooldown_trigger_px_threshold=10, depth_min=0.1, depth_target=10, depth_max=1000,
             # register_mapping.py flags (:63-72)
             inlieralpha=100.0, maxpixelerror=100.0,
             # [additive] 16-bit operand format of encoder and head: "bf16" / "fp16" (the reference's autocast arithmetic); None: $ACEZ_DTYPE, else head.DEFAULT_DTYPE
             compute_dtype=None)
    unknown = set(over) - set(o)
    if unknown:
        raise TypeError(f"unknown options: {sorted(unknown)}")
    o.update(over)
    return SimpleNamespace(**o)


def warp_views(images_b1hw, scale, angles, jitter=None):
    """Augmented views of B normalised grey frames as ONE batched affine warp (dataset.py:283-343): every frame is resized by
    `scale` (canvas int(H*scale) x int(W*scale), shared by the batch) and rotated by its own angle (radians) about its centre --
    content at offset d from the centre moves to R' d with R' = [[cos, sin], [-sin, cos]] in (x right, y down) pixel coordinates,
    which is exactly how camera coordinates transform under aug_pose_inv = pose_rot^-1 (dataset.py:337-343,397) -- bilinear,
    image reflect-padded, mask zero-padded (:327-328). jitter = (brightness [B], contrast [B]) factors of ColorJitter on the
    [0,1] grey values (:148), or None.
    Returns (views [B,1,hs,ws], masks [B,1,hs,ws] bool, grid [B,hs,ws,2] of normalised source coordinates)."""
    B, _, H, W = images_b1hw.shape
    dev = images_b1hw.device
    # output offset p' (pixels, scaled canvas) -> source offset p = R'^T p' / scale; in normalised coordinates x_n = x / (W/2)
    theta, hs, ws = warp_theta(H, W, scale, angles)
    theta = theta.to(dev)
    grid = torch.nn.functional.affine_grid(theta, (B, 1, hs, ws), align_corners=False)
    src = images_b1hw
    if jitter is not None:
        br = torch.as_tensor(jitter[0], dtype=torch.float32).reshape(B, 1, 1, 1).to(dev)
        ct = torch.as_tensor(jitter[1], dtype=torch.float32).reshape(B, 1, 1, 1).to(dev)
        g = ((src * 0.25 + 0.4) * br).clamp(0, 1)
        m = g.mean(dim=(1, 2, 3), keepdim=True)
        g = ((g - m) * ct + m).clamp(0, 1)
        src = (g - 0.4) / 0.25
    views = torch.nn.functional.grid_sample(src, grid, mode="bilinear", padding_mode="reflection", align_corners=False)
    # mask = "a zero-padded bilinear lookup into an all-ones image is positive": at least one of the four taps lies inside the frame with a
    # non-zero weight, i.e. the source coordinate is in the open intervals (-1, W) x (-1, H) -- two comparisons on the grid instead of a
    # second grid_sample over a ones image (same mask, bit for bit, on every view of tests/test_session_cpu.py)
    ix, iy = ((grid[..., 0] + 1) * W - 1) / 2, ((grid[..., 1] + 1) * H - 1) / 2
    masks = ((ix > -1) & (ix < W) & (iy > -1) & (iy < H))[:, None]
    return views, masks, grid


def warp_theta(H, W, scale, angles):
    """The affine maps of warp_views ([B,2,3], normalised output -> normalised source coordinates, align_corners=False) and the canvas size."""
    hs, ws = int(H * scale), int(W * scale)
    sx, sy = ws / W, hs / H
    ang = torch.as_tensor(angles, dtype=torch.float32).reshape(-1)
    c, s_ = torch.cos(ang), torch.sin(ang)
    z = torch.zeros_like(c)
    theta = torch.stack([torch.stack([c * ((ws / 2) / sx / (W / 2)), -s_ * ((hs / 2) / sx / (W / 2)), z], dim=1),
                         torch.stack([s_ * ((ws / 2) / sy / (H / 2)), c * ((hs / 2) / sy / (H / 2)), z], dim=1)], dim=1)
    return theta, hs, ws


def warp_views_device(images_n1hw, image_index, scale, angles, jitter=None, mask_hw=None):
    """warp_views on the GPU through acez_buffer_warp_views (one launch per batch; no sampling grid, no gathered copy of the source frames):
    views of frames `image_index` of the resident table `images_n1hw`. Returns (views [B,1,hs,ws] float32, mask): mask is uint8
    [B,1,oh,ow] at the feature resolution `mask_hw` = (oh, ow) -- the cells the nearest-neighbour resize of ace_trainer.py:373-374 reads -- or
    None. Same arithmetic as warp_views (tests/test_buffer_gpu.py: values to 1e-5, masks identical)."""
    import ctypes as C
    from . import _native as N
    assert images_n1hw.is_cuda and images_n1hw.dtype == torch.float32 and images_n1hw.is_contiguous()
    n, _, H, W = images_n1hw.shape
    dev = images_n1hw.device
    theta, hs, ws = warp_theta(H, W, scale, angles)
    B = theta.shape[0]
    idx = torch.as_tensor(image_index, dtype=torch.int32).reshape(B).to(dev)
    th = theta.reshape(B, 6).contiguous().to(dev)
    jt = None
    if jitter is not None:
        jt = torch.stack([torch.as_tensor(jitter[0], dtype=torch.float32).reshape(B), torch.as_tensor(jitter[1], dtype=torch.float32).reshape(B)], dim=1).contiguous().to(dev)
    views = torch.empty((B, 1, hs, ws), dtype=torch.float32, device=dev)
    mask = torch.empty((B, 1) + tuple(mask_hw), dtype=torch.uint8, device=dev) if mask_hw is not None else None
    scratch = torch.empty((B,), dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    N.check(N.lib().acez_buffer_warp_views(p(images_n1hw), n, H, W, p(idx), p(th), p(jt), B, hs, ws, p(views), p(mask),
                                           mask_hw[0] if mask_hw else 0, mask_hw[1] if mask_hw else 0, p(scratch),
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return views, mask


def warp_view(image_11hw, scale, angle, jitter=None):
    """warp_views for one frame; jitter = (brightness, contrast) floats or None."""
    return warp_views(image_11hw, scale, [angle], None if jitter is None else ([jitter[0]], [jitter[1]]))


def view_depth(depth_11hw, grid, oh, ow):
    """Depth of an augmented view at its feature-map pixel centres (8x+4, 8y+4): nearest lookup in the frame's depth map through
    the view's sampling grid (dataset.py:331-334 resizes / rotates the depth with order=0; depth is invariant to both)."""
    hs, ws = grid.shape[1:3]
    iy = (torch.arange(oh, device=grid.device) * 8 + 4).clamp(max=hs - 1)
    ix = (torch.arange(ow, device=grid.device) * 8 + 4).clamp(max=ws - 1)
    gsub = grid[:, iy][:, :, ix]
    return torch.nn.functional.grid_sample(depth_11hw, gsub, mode="nearest", padding_mode="zeros", align_corners=False)[0, 0]


class ReconstructionSession:
    def __init__(self, encoder_state_dict, images, opt=None, depth=None, device=None, chunk=64, group=None):
        """images [n,1,H,W] float32 normalised (dataset.py:150-153), any device; depth [n,H/8,W/8] camera z at the feature-map
        pixel centres (metres, 0 = invalid) or None -- only the seed images' maps are read.

        Under torch.distributed (one process per GPU, `torchrun ace_zero.py ...`) every rank holds all normalised frames (1.2 MB
        each) but encodes and caches the features of ITS frames only (frame i -> rank i % world, parallel.frames_of_rank):
        registration is sharded by frame with one gather of the poses, a mapping round shards the training buffer by image and
        sums the flat gradient bucket with one all-reduce per step (global batch unchanged), the seed trials run on different
        ranks side by side. Every rank takes the same decisions from the same gathered numbers; rank 0 writes the files."""
        if not torch.cuda.is_available():
            raise RuntimeError("ReconstructionSession needs a GPU: every stage is a HIP kernel (no CPU fallback)")
        self.opt = opt or default_options()
        n, _, H, W = images.shape
        self.n, self.H, self.W = int(n), int(H), int(W)
        check_frame_size(int(H), int(W))                                 # before any frame is encoded
        amax = float(self.opt.aug_scale) if self.opt.use_aug else 1.0
        self.enc = Encoder.from_state_dict(encoder_state_dict, max_frames=chunk, max_h=int(math.ceil(H * amax)) + 8,
                                           max_w=int(math.ceil(W * amax)) + 8, device=device, dtype=getattr(self.opt, "compute_dtype", None))
        self.dtype = self.enc.dtype                                      # one operand format for the encoder, the cached features and every head
        self.dev = self.enc.device
        self.images = images.to(self.dev, torch.float32).contiguous()    # 1.2 MB per 480x640 frame: resident for the augmented passes
        self._aug_rng = np.random.default_rng(self.opt.base_seed + 77)
        self.oh, self.ow = output_size(H, W)
        self.hw = self.oh * self.ow
        self.group = group
        self.rank, self.world = rank_world(group)
        if self.rank:   # data-parallel fills draw their augmentations / sample positions from per-rank streams (rank 0: the single-process one),
            self._aug_rng = np.random.default_rng([self.opt.base_seed + 77, self.rank])   # so the shards' views are not copies of one sequence
        self.owned = np.arange(self.rank, self.n, self.world)            # parallel.frames_of_rank; local slot of frame g: g // world
        self.features = torch.empty((len(self.owned), self.hw, self.enc.out_channels), dtype=self.enc.feature_dtype, device=self.dev)
        t0 = time.time()
        own = torch.from_numpy(self.owned).to(self.dev)
        for c0 in range(0, len(self.owned), chunk):
            c1 = min(len(self.owned), c0 + chunk)
            self.enc.features_rows(self.images[own[c0:c1]], out=self.features[c0:c1].view(-1, self.enc.out_channels))
        torch.cuda.synchronize(self.dev)
        self.depth = None if depth is None else depth.to(self.dev, torch.float32)
        f_ext = float(self.opt.use_external_focal_length)
        self.focal0 = f_ext if f_ext > 0 else math.sqrt(W ** 2 + H ** 2) * 0.7           # dataset.py:269-274
        self.ppx, self.ppy = W / 2.0, H / 2.0                                            # dataset.py:411-412
        self.history = []
        self._views_sampled = 0
        # where a reconstruction's wall-clock goes (reconstruct() returns it): encoding the frames once, buffer creation / training loop of
        # the mapping rounds, registration passes
        self.timings = {"encode_s": time.time() - t0, "buffer_s": 0.0, "loop_s": 0.0, "register_s": 0.0}
        _logger.info(f"Encoded {len(self.owned)} of {self.n} frames in {time.time() - t0:.2f}s (rank {self.rank} of {self.world}); "
                     f"features resident: {self.features.numel() * 2 / 2 ** 30:.2f} GiB")

    def _slots(self, frame_ids):
        """Positions in self.features of frames this rank owns."""
        ids = np.asarray(frame_ids, np.int64)
        if np.any(ids % self.world != self.rank):
            raise ValueError("frame not owned by this rank")
        return torch.from_numpy(ids // self.world).to(self.dev)

    # ------------------------------------------------------------------------------------------------ mapping (train_ace.py)
    def _K(self, focal):
        return torch.tensor([[focal, 0, self.ppx], [0, focal, self.ppy], [0, 0, 1.0]], dtype=torch.float32)

    def _fill_buffer(self, image_ids, poses_c2w, focal, with_depth, total=None):
        """TrainerACE.create_training_buffer (ace_trainer.py:293-452) on cached features, views not augmented."""
        o = self.opt
        ids = torch.as_tensor(list(image_ids), dtype=torch.long, device=self.dev)
        m = len(ids)
        total = total if total is not None else min(o.max_training_buffer_size, o.max_dataset_passes * m * o.samples_per_image)
        C = self.enc.out_channels
        feats = torch.empty((total, C), dtype=self.enc.feature_dtype, device=self.dev)
        px = torch.empty((total, 2), dtype=torch.float32, device=self.dev)
        vidx = torch.empty((total,), dtype=torch.int32, device=self.dev)
        pix = torch.empty((total,), dtype=torch.int32, device=self.dev)
        src = self.features[self._slots(ids.cpu().numpy())].reshape(-1, C)   # one copy of the mapped images' feature maps
        mask = None
        if with_depth:                                                  # pixels without a depth are not sampled (dataset.py:384-386 zero them)
            d = self.depth[ids]
            mask = ((d > 0) & (d <= 1000)).to(torch.uint8).reshape(m, self.hw).contiguous()
        filled, n_views = 0, 0
        while filled < total:
            v = min(m, (total - filled + o.samples_per_image - 1) // o.samples_per_image)
            take = min(v * o.samples_per_image, total - filled)
            if take < v * o.samples_per_image:                           # last, partial view: sample whole views into scratch
                of = torch.empty((v * o.samples_per_image, C), dtype=self.enc.feature_dtype, device=self.dev)
                op = torch.empty((v * o.samples_per_image, 2), dtype=torch.float32, device=self.dev)
                ov = torch.empty((v * o.samples_per_image,), dtype=torch.int32, device=self.dev)
                ox = torch.empty((v * o.samples_per_image,), dtype=torch.int32, device=self.dev)
            else:
                of, op, ov, ox = (t[filled:filled + take] for t in (feats, px, vidx, pix))
            N.check(N.lib().acez_buffer_sample_views(_ptr(src), _ptr(mask) if mask is not None else None, v, self.oh, self.ow, C,
                                                     o.samples_per_image, o.base_seed + 4095 + 7919 * self.rank, self._views_sampled, n_views, _ptr(of), _ptr(op),
                                                     _ptr(ov), _ptr(ox), _stream()))
            if take < v * o.samples_per_image:
                feats[filled:], px[filled:], vidx[filled:], pix[filled:] = of[:take], op[:take], ov[:take], ox[:take]
            filled += take
            n_views += v
            self._views_sampled += v
        view_image = torch.arange(n_views, dtype=torch.int32) % m        # pass p, view j -> image j
        poses_c2w = torch.as_tensor(poses_c2w, dtype=torch.float64).reshape(m, 4, 4)
        pose_inv = torch.linalg.inv(poses_c2w).to(torch.float32)
        K = self._K(focal)
        buf = dict(features=feats, target_px=px, view_idx=vidx, view_aug_inv=torch.eye(4)[:3].repeat(n_views, 1, 1),
                   view_K=K.repeat(n_views, 1, 1), view_Kinv=torch.linalg.inv(K).repeat(n_views, 1, 1), view_image=view_image,
                   image_pose_inv=pose_inv)
        if with_depth:
            # ground-truth scene coordinates from depth (dataset.py:347-388): eye = ((x*8+4 - W/2) / f * d, (y*8+4 - H/2) / f * d, d)
            img = (vidx.long() % m)
            d = self.depth[ids].reshape(m, self.hw)[img, pix.long()]
            eye = torch.stack([(px[:, 0] - self.ppx) / focal * d, (px[:, 1] - self.ppy) / focal * d, d, torch.ones_like(d)], dim=1)
            buf["target_crds"] = torch.einsum("nij,nj->ni", poses_c2w.to(self.dev, torch.float32)[img][:, :3], eye)
        return buf

    AUG_SCALE_LEVELS = 16

    def _draw_augmentations(self, m):
        """Per view: scale, rotation angle (radians), brightness, contrast (dataset.py:420-427,324,148). The scale is drawn from
        AUG_SCALE_LEVELS equally spaced values of [1/s, s] instead of the continuum, so that the views of a pass fall into a few
        canvas sizes and go through warp / encoder / sampling in batches (one view at a time the fill is host-bound: 0.6 ms per
        view against 0.09 ms of encoder time)."""
        o, rng = self.opt, self._aug_rng
        lo, hi = 1.0 / o.aug_scale, o.aug_scale
        levels = rng.integers(0, self.AUG_SCALE_LEVELS, size=m)
        scales = lo + (levels + 0.5) * (hi - lo) / self.AUG_SCALE_LEVELS
        angles = np.radians(rng.uniform(-o.aug_rotation, o.aug_rotation, size=m))
        bw = float(o.aug_black_white)
        jit = (rng.uniform(1 - bw, 1 + bw, size=m), rng.uniform(1 - bw, 1 + bw, size=m)) if bw > 0 else None
        return levels, scales, angles, jit

    @staticmethod
    def _rot_inv(angles):
        """Inverse of pose_rot (dataset.py:337-343) for a batch of angles: [[c, s], [-s, c]] in the upper-left block."""
        a = torch.as_tensor(angles, dtype=torch.float32)
        r = torch.eye(4).repeat(len(a), 1, 1)
        r[:, 0, 0], r[:, 0, 1], r[:, 1, 0], r[:, 1, 1] = torch.cos(a), torch.sin(a), -torch.sin(a), torch.cos(a)
        return r

    def _fill_buffer_augmented(self, image_ids, poses_c2w, focal, with_depth, total=None):
        """create_training_buffer with --use_aug True: every pass re-encodes a freshly augmented view of every mapped image and
        samples it with its validity mask. The views of a pass are grouped by canvas size and processed in batches."""
        from .buffer import BufferBuilder
        o = self.opt
        ids = torch.as_tensor([int(i) for i in image_ids], dtype=torch.long)
        m = len(ids)
        total = total if total is not None else min(o.max_training_buffer_size, o.max_dataset_passes * m * o.samples_per_image)
        bld = BufferBuilder(self.enc, capacity=total, samples_per_image=o.samples_per_image, seed=o.base_seed + 4095 + 7919 * self.rank + self._views_sampled)
        poses_c2w = torch.as_tensor(poses_c2w, dtype=torch.float64).reshape(m, 4, 4)
        pose_inv = torch.linalg.inv(poses_c2w).to(torch.float32)
        # views per warp / encoder / sampling batch (<= the encoder's max_frames). 64 (round 6; 32 before, 16 in round 3): the 3 x 3 patch
        # kernel gets two to five waves of tiles per layer instead of one or two (1000-frame session, 16 scale levels: buffer creation
        # 6.71 -> 5.26 s of 26.6 s; 128 views per batch or 8 scale levels add nothing: profiles/r06_session_chunk_sweep.log)
        chunk = int(os.environ.get("ACEZ_AUG_CHUNK", "64"))
        chunk = max(1, min(chunk, self.enc.max_frames))
        crds = []
        while not bld.full:
            rows_before = bld.n
            levels, scales, angles, jit = self._draw_augmentations(m)
            for lv in np.unique(levels):
                sel = np.flatnonzero(levels == lv)
                scale = float(scales[sel[0]])
                hs, ws = int(self.H * scale), int(self.W * scale)
                f = focal * (hs / self.H)                                # dataset.py:289-290 scales the focal with the short side
                K = torch.tensor([[f, 0, ws / 2.0], [0, f, hs / 2.0], [0, 0, 1.0]])
                Kinv = torch.linalg.inv(K)
                for c0 in range(0, len(sel), chunk):
                    if bld.full:
                        break
                    js = sel[c0:c0 + chunk]
                    b = len(js)
                    rot_inv = self._rot_inv(angles[js])
                    if not with_depth:
                        # one launch: warp + colour jitter + the mask at feature resolution (acez_buffer_warp_views); a rotation of at most
                        # aug_rotation degrees never empties the mask: no per-batch host synchronisation
                        views, mask8 = warp_views_device(self.images, ids[js], scale, angles[js], None if jit is None else (jit[0][js], jit[1][js]),
                                                         mask_hw=output_size(hs, ws))
                        bld.add_views(views, mask8, rot_inv, pose_inv[js], K.repeat(b, 1, 1), Kinv.repeat(b, 1, 1), [int(j) for j in js],
                                      check_empty=False, mask_at_feature_resolution=True)
                        continue
                    views, masks, grid = warp_views(self.images[ids[js].to(self.dev)], scale, angles[js], None if jit is None else (jit[0][js], jit[1][js]))
                    # depth-supervised (seed) views: depth at the feature-map pixel centres through the same warp (dataset.py:331-334, order=0)
                    oh, ow = output_size(hs, ws)
                    for k, j in enumerate(js):
                        n0 = bld.n
                        dv = view_depth(self.depth[int(ids[j])][None, None], grid[k:k + 1], oh, ow)
                        valid = (dv > 0) & (dv <= 1000)
                        mk = torch.nn.functional.interpolate(masks[k:k + 1].float(), size=(oh, ow), mode="nearest")[0, 0] > 0
                        took = bld.add_views(views[k:k + 1], (mk & valid).float()[None, None], rot_inv[k:k + 1], pose_inv[j:j + 1], K[None], Kinv[None],
                                             [int(j)], want_pixels=True, mask_at_feature_resolution=True)
                        if took:
                            pix = bld.last_pixels[:took].long()
                            dd = dv[pix // ow, pix % ow]
                            px = bld.target_px[n0:n0 + took]
                            eye = torch.stack([(px[:, 0] - ws / 2.0) / f * dd, (px[:, 1] - hs / 2.0) / f * dd, dd, torch.ones_like(dd)], dim=1)
                            T = (poses_c2w[j].to(torch.float32) @ torch.linalg.inv(rot_inv[k])).to(self.dev)   # pose @ pose_rot (dataset.py:381)
                            xyz = eye @ T[:3].T
                            xyz[(dd <= 0) | (dd > 1000)] = 0                 # "unavailable" marker (dataset.py:384-386); unreachable through the mask
                            crds.append(xyz)
            if with_depth and bld.n == rows_before:
                # the reference samples without looking at the depth and cannot stall; here depth-less cells are masked out, so a seed
                # image without any usable depth would never fill the buffer
                raise RuntimeError("no training sample in a whole pass over images %s: their depth maps have no value in (0, 1000] inside "
                                   "the augmented views (all zeros, wrong units, or a depth file that belongs to another image)"
                                   % [int(i) for i in ids.tolist()])
        self._views_sampled += bld.n_views
        buf = bld.finish()
        if with_depth:
            buf["target_crds"] = torch.cat(crds)[:int(buf["features"].shape[0])]
        return buf

    def map(self, image_ids, poses_c2w, focal, *, iterations, loss_type, schedule, lr_max, refinement="none", pose_wait=0,
            refine_calibration=False, load_weights=None, with_depth=False, tag="map", data_parallel=None):
        """One train_ace.py run (ace_trainer.py:TrainerACE.train): returns {"head": fp16 state_dict, "poses_w2c": [m,3,4] refined,
        "focal": refined focal, "iterations", "seconds", "patches_per_s", "batch_inliers"}.

        data_parallel (default: whenever the process group has more than one rank and there are at least as many images as ranks):
        the mapped images are dealt to the ranks, every rank fills its share of the buffer from its images, and every step is
        backward -> all-reduce of the flat gradient bucket -> update with the reference's batch composition: all ranks draw the same
        permutation of the GLOBAL buffer and take the rows of each 5120-slice that live in their shard (parallel.epoch_local_batches;
        the loss is a sum / 5120, ace_trainer.py:612-613, so the summed gradient is the single-GPU gradient of that batch). Replicas
        stay bit-identical without a weight broadcast, so every rank returns the same result."""
        o = self.opt
        t0 = time.time()
        image_ids = [int(i) for i in image_ids]
        m = len(image_ids)
        if data_parallel is None and self.world > 1 and m < self.world:
            # Fewer images than ranks: nothing to shard. ONE rank trains and hands its result to the others -- if every rank trained its
            # own copy, their augmentation / sampling streams (advanced by different amounts in earlier calls) would give different
            # heads, and the next warm-started data-parallel round would all-reduce gradients taken at different weights.
            owners = sorted({i % self.world for i in image_ids})
            if not o.use_aug and len(owners) != 1:
                raise RuntimeError("mapping fewer images than ranks without augmentation needs all of them on one rank (cached features "
                                   "live on the frame's owner): use --use_aug True or fewer GPUs")      # (the same decision on every rank)
            src = 0 if o.use_aug else owners[0]
            box = [self.map(image_ids, poses_c2w, focal, iterations=iterations, loss_type=loss_type, schedule=schedule, lr_max=lr_max,
                            refinement=refinement, pose_wait=pose_wait, refine_calibration=refine_calibration, load_weights=load_weights,
                            with_depth=with_depth, tag=tag, data_parallel=False) if self.rank == src else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)
            return box[0]
        poses_c2w = torch.as_tensor(poses_c2w, dtype=torch.float32).reshape(-1, 4, 4)
        dp = (self.world > 1 and m >= self.world) if data_parallel is None else bool(data_parallel and self.world > 1)
        fill = self._fill_buffer_augmented if o.use_aug else self._fill_buffer
        total = min(o.max_training_buffer_size, o.max_dataset_passes * m * o.samples_per_image)
        if dp:
            # augmented fill re-encodes every view, so any rank can take any image: deal them by position; the plain fill reads the
            # cached feature maps, which live on the frame's owner
            owner = [(j if o.use_aug else image_ids[j]) % self.world for j in range(m)]
            counts = [owner.count(r) for r in range(self.world)]
            if min(counts) == 0:
                raise RuntimeError("data-parallel mapping: a rank owns none of the mapped images (use --use_aug True or fewer GPUs)")
            shares = [total * c // m for c in counts]
            for r in range(total - sum(shares)):
                shares[r % self.world] += 1
            mine = [j for j in range(m) if owner[j] == self.rank]
            buf = fill([image_ids[j] for j in mine], poses_c2w[mine], focal, with_depth, total=shares[self.rank])
            # per-image table: ALL mapped images on every rank (the pose network / the per-image pose parameters are replicated and
            # their gradient is part of the all-reduced bucket); the views of this shard point into it by global position
            buf["view_image"] = torch.as_tensor(mine, dtype=torch.int32)[buf["view_image"].long()]
            buf["image_pose_inv"] = torch.linalg.inv(poses_c2w.double()).float()
            shard_lo = sum(shares[:self.rank])
            n = total
        else:
            buf = fill(image_ids, poses_c2w, focal, with_depth)
            shard_lo, n = 0, int(buf["features"].shape[0])
        torch.cuda.synchronize(self.dev)
        t_fill = time.time() - t0
        n_local = int(buf["features"].shape[0])
        if dp and n_local != shares[self.rank]:
            raise RuntimeError(f"rank {self.rank} filled {n_local} buffer rows instead of its share of {shares[self.rank]}")
        if n < o.batch_size:
            raise ValueError(f"training buffer of {n} patches is smaller than one batch ({o.batch_size})")
        if load_weights is not None:
            mean = load_weights["mean"].float().view(3)                  # Regressor.create_from_split_state_dict keeps the stored mean
        else:
            mean = poses_c2w[:, :3, 3].mean(dim=0)                       # dataset.py:206-225
        tr = HeadTrainer(mean, num_head_blocks=o.num_head_blocks, use_homogeneous=o.use_homogeneous, max_batch=o.batch_size, loss_type=loss_type,
                         soft_clamp=o.repro_loss_soft_clamp, soft_clamp_min=o.repro_loss_soft_clamp_min,
                         circle_schedule=o.repro_loss_schedule == "circle", hard_clamp=o.repro_loss_hard_clamp, depth_min=o.depth_min,
                         depth_max=o.depth_max, depth_target=o.depth_target,
                         inlier_px_threshold=o.learning_rate_cooldown_trigger_px_threshold, schedule=schedule, iterations=iterations,
                         lr_min=o.learning_rate_min, lr_max=lr_max,
                         warmup_iterations=o.learning_rate_warmup_iterations, warmup_lr=o.learning_rate_warmup_learning_rate,
                         cooldown_iterations=o.cooldown_iterations, cooldown_trigger_percent=o.cooldown_threshold,
                         refine_calibration=refine_calibration, focal_init=focal, calib_lr=o.refine_calibration_lr, pose_refinement=refinement,
                         pose_refinement_wait=pose_wait, pose_refinement_lr=o.pose_refinement_lr, pose_refinement_weight=o.pose_refinement_weight,
                         refinement_ortho=o.refinement_ortho, pose_seed=o.base_seed + 511,
                         initial_poses=buf["image_pose_inv"][:, :3] if refinement == "naive" else None, device=self.dev.index, dtype=self.dtype)
        if load_weights is not None:
            tr.load_state_dict(load_weights)
        else:
            g = torch.Generator().manual_seed(o.base_seed + 1023)        # ace_trainer.py:66-69
            tr.load_flat((torch.rand(tr.n_params, generator=g) * 2 - 1) / math.sqrt(512.0))
        tr.set_buffer(**buf)
        torch.cuda.synchronize(self.dev)
        t_loop0 = time.time()
        launched, done = 0, False
        perms = epoch_permutations(n, o.base_seed + 8191, self.dev) if dp else None       # ace_trainer.py:79-80 seed of the training generator
        pairs = None if dp else epoch_batches(n, o.batch_size, o.base_seed + 8191, self.dev)  # (rows, next rows): the next batch is gathered ahead
        dpt = make_data_parallel(tr, self.group) if dp else None          # one all-reduce of the gradient bucket per step (parallel.py)
        while not done:                                                  # TrainerACE.train / run_epoch (ace_trainer.py:454-497)
            if dp:
                perm = next(perms)
                local, offs = epoch_local_batches(perm, o.batch_size, shard_lo, shard_lo + n_local)
            nb = n // o.batch_size
            for b in range(nb):
                if dp:
                    # RCCL: head + pose gradients, loss / inlier / focal statistics; the rank's rows of the next batch are announced to the
                    # update (gathered inside the optimiser's launch; the first batch of an epoch gathers for itself)
                    dpt.step(local[offs[b]:offs[b + 1]], local[offs[b + 1]:offs[b + 2]] if b + 1 < nb else None)
                else:
                    tr.step(*next(pairs))
                launched += 1
                if launched % 64 == 0:                                   # the only host synchronisation of the loop
                    st = tr.state()
                    if st["nan"]:
                        raise RuntimeError("Aborting because of NaN loss")   # ace_trainer.py:615-617
                    if st["iteration"] >= st["max_iterations"]:
                        done = True
                        break
        st = tr.state()
        if dpt is not None:
            dpt.gather_masters()                                         # the fp32 masters of the other ranks' layers, for the checkpoint
        dt = time.time() - t0
        t_loop = time.time() - t_loop0
        self.timings["buffer_s"] += t_fill
        self.timings["loop_s"] += t_loop
        out = {"head": {k: v.detach().cpu().half() for k, v in tr.state_dict().items()},      # save_model (ace_trainer.py:681-694)
               "poses_w2c": tr.current_poses(), "focal": float(st["focal_scale"] * focal) if refine_calibration else float(focal),
               "iterations": int(st["iteration"]), "seconds": dt, "fill_seconds": t_fill, "loop_seconds": t_loop,
               "patches_per_s": st["iteration"] * o.batch_size / t_loop,
               "batch_inliers": float(st["batch_inliers"]), "loss": float(st["loss"]), "buffer": n, "data_parallel": bool(dp)}
        _logger.info(f"[{tag}] {m} images, {n} patches{' over %d ranks' % self.world if dp else ''}, {out['iterations']} iterations in {dt:.2f}s "
                     f"(buffer {t_fill:.2f}s, loop {t_loop:.2f}s), batch inliers {out['batch_inliers'] * 100:.1f}%, focal {out['focal']:.1f}")
        tr.close()
        return out

    # ---------------------------------------------------------------------------------------- registration (register_mapping.py)
    def scene_coordinates(self, head_sd, frame_ids=None):
        """Head.forward on the cached features of the given frames (default: every frame this rank owns): float32 [k,3,oh,ow] on
        the device."""
        ids = self.owned if frame_ids is None else np.asarray(frame_ids, np.int64)
        count = len(ids)
        out = torch.empty((count, 3, self.oh, self.ow), dtype=torch.float32, device=self.dev)
        if count == 0:
            return out
        slots = self._slots(ids)
        nb = sum(1 for k in head_sd if k.endswith("c0.weight"))
        head = HeadTrainer(head_sd["mean"].float().view(3), num_head_blocks=nb, use_homogeneous=head_sd["fc3.weight"].shape[0] == 4,
                           max_batch=min(count, 64) * self.hw, iterations=1, inference_only=True, device=self.dev.index,
                           dtype=self.dtype)   # self.features holds the encoder's 16-bit rows, handed over by raw pointer below
        head.load_state_dict(head_sd)
        contiguous = bool(np.all(np.diff(slots.cpu().numpy()) == 1)) if count > 1 else True
        for c0 in range(0, count, 64):
            c1 = min(count, c0 + 64)
            rows = (self.features[int(slots[c0]):int(slots[c0]) + (c1 - c0)] if contiguous else self.features[slots[c0:c1]]).reshape(-1, self.enc.out_channels)
            N.check(head.lib.acez_head_forward_maps(head._h, _ptr(rows), c1 - c0, self.oh, self.ow, _ptr(out[c0:c1]), _stream()))
        torch.cuda.synchronize(self.dev)
        head.close()
        return out

    def register(self, head_sd, focal, max_estimates=-1, tag="register", max_tries=16, rng_ids=None):
        """register_mapping.py:201-276: (poses cam->world [k,4,4] float32, inlier counts [k] int32); the k frame ids they belong to
        are left in self.registered_ids (all frames in order unless max_estimates draws a subset).

        max_estimates > 0 scores a random subset like the reference does: register_mapping.py iterates a DataLoader(shuffle=True)
        under torch.manual_seed(base_seed) and stops after max_estimates frames (:122-147,256) -- a seeded permutation of the frame
        ids here (a different stream, the same law), NOT the first k frames of the sequence. Frames are sharded over the ranks
        (frame i -> rank i % world); the random stream of a frame is keyed by its id, so the result does not depend on the
        partition, and one gather returns every frame's result to every rank. rng_ids: the ids that key the random streams, if
        they are not the positions in this session (register_mapping.py on a folder of mixed frame sizes: one session per size
        class, streams keyed by the position in the whole file list)."""
        o = self.opt
        if max_estimates <= 0 or max_estimates >= self.n:
            ids = np.arange(self.n)
        else:
            g = torch.Generator().manual_seed(int(o.register_seed))
            ids = np.sort(torch.randperm(self.n, generator=g)[:max_estimates].numpy())
        self.registered_ids = ids
        t0 = time.time()
        mine = ids[ids % self.world == self.rank]
        sc = self.scene_coordinates(head_sd, mine)
        prm = dict(hyps=o.ransac_iterations, thr=o.ransac_threshold, alpha=float(o.inlieralpha), max_reproj=float(o.maxpixelerror), sub=8, max_tries=max_tries)
        if len(mine):
            keys = [int(i) for i in mine] if rng_ids is None else [int(rng_ids[i]) for i in mine]
            poses, inl, _ = dsacstar.register_batch(sc, [(focal, self.ppx, self.ppy)] * len(mine), prm, o.register_seed, keys, want_masks=False)
            poses, inl = poses.cpu(), inl.cpu().to(torch.int32)
        else:
            poses, inl = torch.zeros(0, 4, 4), torch.zeros(0, dtype=torch.int32)
        if self.world > 1:
            full_p, full_i = gather_registrations([int(i) for i in mine], poses, inl, self.n, self.group, expect=ids)
            poses, inl = full_p[ids], full_i[ids]
        poses, inl = poses.numpy(), inl.numpy()
        rate = float((inl > o.registration_confidence).mean())
        self.timings["register_s"] += time.time() - t0
        _logger.info(f"[{tag}] {len(ids)} frames in {time.time() - t0:.2f}s, {rate * 100:.1f}% above confidence {o.registration_confidence}")
        return poses, inl

    # --------------------------------------------------------------------------------------------------- the loop (ace_zero.py)
    def map_seed(self, seed_idx, seed):
        """ace_zero_util.map_seed: one image, identity pose, depth-supervised, scored on (at most) 1000 frames."""
        o = self.opt
        img = int(seed * self.n)                                         # dataset.py:112
        if self.depth is None:
            raise RuntimeError("a seed image needs a depth map (ace_zero.py --depth_files); the ZoeDepth fallback is a network download")
        rows = self.opt.max_dataset_passes * self.opt.samples_per_image
        if rows < self.opt.batch_size:                                   # one image gives passes x samples rows (10 240 by default = two batches)
            raise ValueError(f"a seed image yields {rows} buffer rows, fewer than one batch of {self.opt.batch_size}: raise --max_dataset_passes")
        return self.map([img], torch.eye(4).unsqueeze(0), self.focal0, iterations=o.seed_iterations, loss_type=o.repro_loss_type,
                        schedule=o.learning_rate_schedule, lr_max=o.learning_rate_max, with_depth=True, tag=f"iteration0_seed{seed_idx}",
                        data_parallel=False)

    def score_seed(self, seed_idx, m):
        _, inl = self.register(m["head"], self.focal0, max_estimates=self.opt.max_estimates_seed_scoring, tag=f"iteration0_seed{seed_idx}_fastcheck")
        return float((inl > self.opt.registration_confidence).mean())

    def reconstruct(self):
        o = self.opt
        t_start = time.time()
        focal = self.focal0
        if o.seed_network is not None:                                   # ace_zero.py:176-178: a pre-trained head (state_dict) as the seed
            current, first_id, seed_rates = {"head": o.seed_network}, "seed_network", []
        else:
            np.random.seed(o.random_seed)                                # ace_zero.py:181-183
            seeds = np.random.uniform(size=o.try_seeds)
            if self.world == 1:
                trials = []
                for i, sd_ in enumerate(seeds):                          # ace_zero.py:185-215: map, then score, seed after seed
                    mp_ = self.map_seed(i, sd_)
                    trials.append((mp_, self.score_seed(i, mp_)))
            else:
                # a seed trial maps ONE image (nothing to shard): trial i runs on rank i % world, side by side with the others; its
                # head is then handed to every rank and scored by all of them together (sharded registration)
                # (without augmentation the buffer is filled from the cached feature maps, which only the frame's owner holds)
                runner = [(i if o.use_aug else int(sd_ * self.n)) % self.world for i, sd_ in enumerate(seeds)]
                maps = [self.map_seed(i, sd_) if runner[i] == self.rank else None for i, sd_ in enumerate(seeds)]
                for i in range(len(maps)):
                    box = [maps[i]]
                    dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, runner[i]) if self.group is not None else runner[i],
                                               group=self.group)
                    maps[i] = box[0]
                trials = [(mp_, self.score_seed(i, mp_)) for i, mp_ in enumerate(maps)]
            best = int(np.argmax([r for _, r in trials]))
            current, first_id, seed_rates = trials[best][0], f"iteration0_seed{best}", [r for _, r in trials]
        poses, conf = self.register(current["head"], focal, tag=first_id)
        max_rate = float((conf > o.registration_confidence).mean())
        self.history.append({"id": first_id, "registration_rate": max_rate, "focal": focal, "seed_rates": seed_rates, "poses": poses,
                             "confidence": conf, "head": current["head"]})
        scheduled_to_stop_early = False
        iteration = 0
        for iteration in range(1, o.iterations_max):
            sel = np.flatnonzero(conf >= o.registration_confidence)      # --ace_pose_file_conf_threshold: load_dataset_ace drops `confidence < threshold`
            if len(sel) == 0:
                raise RuntimeError("no image is registered above the confidence threshold: cannot continue the reconstruction")
            refit = scheduled_to_stop_early and o.final_refit
            warm = o.warmstart and (iteration > 1 or o.seed_network is not None) and not refit   # ace_zero.py:266-270
            if refit:                                                    # ace_zero_util.get_refit_mapping_cmd
                kw = dict(iterations=o.refit_iterations, loss_type="dyntanh", schedule="circle", lr_max=0.005, pose_wait=o.final_refit_posewait)
            else:                                                        # ace_zero_util.get_base_mapping_cmd
                kw = dict(iterations=o.iterations, loss_type=o.repro_loss_type, schedule=o.learning_rate_schedule, lr_max=o.learning_rate_max,
                          pose_wait=o.pose_refinement_wait)
            current = self.map(sel, torch.from_numpy(poses[sel]), focal, refinement=o.refinement, refine_calibration=o.refine_calibration,
                               load_weights=current["head"] if warm else None, tag=f"iteration{iteration}", **kw)
            focal = current["focal"]                                     # ace_zero.py:297-305: the refined focal goes to the registration
            poses, conf = self.register(current["head"], focal, tag=f"iteration{iteration}")
            rate = float((conf > o.registration_confidence).mean())
            self.history.append({"id": f"iteration{iteration}", "registration_rate": rate, "focal": focal, "mapped_images": int(len(sel)),
                                 "iterations": current["iterations"], "map_seconds": current["seconds"], "refit": bool(refit), "poses": poses,
                                 "confidence": conf, "head": current["head"], "poses_w2c_refined": current["poses_w2c"]})
            if scheduled_to_stop_early:
                break
            if rate >= o.registration_threshold or (rate - max_rate) < o.relative_registration_threshold:   # ace_zero.py:317-327
                if o.final_refine:
                    scheduled_to_stop_early = True
                else:
                    break
            if iteration >= o.iterations_max - 2:
                scheduled_to_stop_early = True
            max_rate = max(rate, max_rate)
        out = {"poses": poses, "confidence": conf, "focal": focal, "head": current["head"], "history": self.history,
               "iterations": iteration, "seconds": time.time() - t_start, "timings": dict(getattr(self, "timings", {}))}
        if o.export_point_cloud:                                         # ace_zero.py:379-400
            out["point_cloud"] = self.point_cloud(current["head"], poses, conf, focal, dense=o.dense_point_cloud)
        return out

    def point_cloud(self, head_sd, poses_c2w, confidence, focal, dense=False, filter_depth=100, opengl=False):
        """export_point_cloud.py:66-94 on the cached features of the registered frames (confidence > registration_confidence):
        (xyz [N,3] float32, source [N] = position in the registered list * hw + map pixel). OpenCV convention by default, as
        ace_zero.py requests it (--convention opencv, :398)."""
        from .pointcloud import filter_scene_coordinates
        sel = np.flatnonzero(np.asarray(confidence) >= self.opt.registration_confidence)   # load_dataset_ace keeps confidence >= threshold
        if self.world > 1:
            raise NotImplementedError("point-cloud export runs on one GPU (python export_point_cloud.py on the written pose file)")
        sc = self.scene_coordinates(head_sd, sel)
        pinv = torch.linalg.inv(torch.from_numpy(np.asarray(poses_c2w, np.float64)[sel])).to(torch.float32)
        K = self._K(focal).repeat(len(sel), 1, 1)
        xyz, src, _, _ = filter_scene_coordinates(sc, pinv, K, filter_depth, dense, len(sel), seed=self.opt.random_seed, opengl=opengl)
        return xyz.cpu().numpy(), src.cpu().numpy(), sel


def write_pose_file(path, image_names, poses_c2w, confidences, focal):
    """poses_<session>.txt (register_mapping.py:261-276): world->camera quaternion + translation, focal, confidence."""
    from .cli import write_pose_line
    with open(path, "w") as f:
        for name, p, c in zip(image_names, poses_c2w, confidences):
            write_pose_line(f, name, np.linalg.inv(np.asarray(p, np.float64)), int(c), float(focal))

"""Training-buffer creation on the device: mirrors TrainerACE.create_training_buffer
(ace_trainer.py:293-452) from the point where a batch of augmented views reaches the GPU.

    bld = BufferBuilder(encoder, capacity=8_000_000, samples_per_image=1024, seed=2089)
    bld.add_views(image_V1HW, mask_V1HW, aug_pose_inv_V44, pose_inv_V44, K_V33, Kinv_V33, image_index_V)   # per dataloader batch
    buf = bld.finish()        # dict for HeadTrainer.set_buffer(**buf)

What differs from the reference, by design: per-view data (augmentation pose, intrinsics, image index) is
stored once per view and per-image poses once per image instead of once per patch; features are 16-bit in the encoder's operand format (bf16, or fp16 as the reference stores them, ace_trainer.py:330). The image pipeline that
produces the views (decode / resize / rotate / jitter, dataset.py) is not part of this package.
There is no CPU fallback.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _native as N
from .encoder import output_size


class BufferBuilder:
    def __init__(self, encoder, capacity, samples_per_image=1024, seed=2089, n_images=None):
        self.enc = encoder
        self.dev = encoder.device
        self.capacity = int(capacity)               # --max_training_buffer_size (train_ace.py:122)
        self.samples = int(samples_per_image)       # --samples_per_image (train_ace.py:128)
        self.seed = int(seed)
        self.n = 0
        self.features = torch.empty((self.capacity, encoder.out_channels), dtype=encoder.feature_dtype, device=self.dev)
        self.target_px = torch.empty((self.capacity, 2), dtype=torch.float32, device=self.dev)
        self.view_idx = torch.empty((self.capacity,), dtype=torch.int32, device=self.dev)
        self._aug, self._K, self._Kinv, self._img = [], [], [], []
        self._pose_inv = {}
        self.n_views = 0

    @property
    def full(self):
        return self.n >= self.capacity

    def add_views(self, image_v1hw, mask_v1hw, aug_pose_inv_v44, pose_inv_v44, K_v33, Kinv_v33, image_index_v, want_pixels=False, check_empty=True,
                  mask_at_feature_resolution=False):
        """One dataloader batch (the reference uses batch size 1; any number of same-sized views works here).
        mask_at_feature_resolution: mask_v1hw is already [v,1,oh,ow] (a mask combined with per-cell data such as depth validity): it is
        used as is -- up- and down-sampling it through the image resolution is not the identity when h or w is not a multiple of 8."""
        if self.full:
            return 0
        v, _, h, w = image_v1hw.shape
        oh, ow = output_size(h, w)
        # mask at feature resolution, nearest neighbour (ace_trainer.py:373-374)
        if mask_v1hw is not None:
            if mask_at_feature_resolution:
                assert tuple(mask_v1hw.shape[-2:]) == (oh, ow), (tuple(mask_v1hw.shape), (oh, ow))
                m = mask_v1hw.to(self.dev) > 0
            else:
                m = F.interpolate(mask_v1hw.to(self.dev, torch.float32), size=(oh, ow), mode="nearest") > 0
            # views without a valid pixel are skipped (ace_trainer.py:377-378); the test costs a host synchronisation per call,
            # callers that know their masks are never empty (rotations of a few degrees) switch it off
            keep = m.flatten(1).any(dim=1) if check_empty else None
            if check_empty and not bool(keep.all()):
                sel = keep.nonzero().flatten().cpu()
                if len(sel) == 0:
                    return 0
                return self.add_views(image_v1hw[sel], mask_v1hw[sel], aug_pose_inv_v44[sel], pose_inv_v44[sel], K_v33[sel],
                                      Kinv_v33[sel], [image_index_v[int(i)] for i in sel], want_pixels=want_pixels, check_empty=False,
                                      mask_at_feature_resolution=mask_at_feature_resolution)
            mask_u8 = m.to(torch.uint8).contiguous()
        else:
            mask_u8 = None
        # the last batch is truncated at the capacity (ace_trainer.py:415-416)
        room_views = (self.capacity - self.n + self.samples - 1) // self.samples
        if v > room_views:
            v = room_views
            image_v1hw, aug_pose_inv_v44, pose_inv_v44 = image_v1hw[:v], aug_pose_inv_v44[:v], pose_inv_v44[:v]
            K_v33, Kinv_v33, image_index_v = K_v33[:v], Kinv_v33[:v], list(image_index_v)[:v]
            mask_u8 = mask_u8[:v].contiguous() if mask_u8 is not None else None
        rows = self.enc.features_rows(image_v1hw)
        n_new = v * self.samples
        take = min(n_new, self.capacity - self.n)
        if take < n_new:   # partial last view: sample into scratch, copy what fits
            of = torch.empty((n_new, self.enc.out_channels), dtype=self.enc.feature_dtype, device=self.dev)
            op = torch.empty((n_new, 2), dtype=torch.float32, device=self.dev)
            ov = torch.empty((n_new,), dtype=torch.int32, device=self.dev)
        else:
            of, op, ov = self.features[self.n:self.n + n_new], self.target_px[self.n:self.n + n_new], self.view_idx[self.n:self.n + n_new]
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        self.last_pixels = torch.empty((n_new,), dtype=torch.int32, device=self.dev) if want_pixels else None   # chosen map pixels (y * ow + x)
        N.check(N.lib().acez_buffer_sample_views(C.c_void_p(rows.data_ptr()), C.c_void_p(mask_u8.data_ptr()) if mask_u8 is not None else None,
                                                 v, oh, ow, self.enc.out_channels, self.samples, C.c_uint64(self.seed),
                                                 C.c_uint64(self.n_views), self.n_views, C.c_void_p(of.data_ptr()),
                                                 C.c_void_p(op.data_ptr()), C.c_void_p(ov.data_ptr()),
                                                 C.c_void_p(self.last_pixels.data_ptr()) if want_pixels else None, stream))
        if take < n_new:
            self.features[self.n:self.n + take] = of[:take]
            self.target_px[self.n:self.n + take] = op[:take]
            self.view_idx[self.n:self.n + take] = ov[:take]
        self.n += take
        self.n_views += v
        self._aug.append(aug_pose_inv_v44[:, :3].to(torch.float32).cpu())
        self._K.append(K_v33.to(torch.float32).cpu())
        self._Kinv.append(Kinv_v33.to(torch.float32).cpu())
        for i in range(v):
            idx = int(image_index_v[i])
            self._img.append(idx)
            self._pose_inv[idx] = pose_inv_v44[i].to(torch.float32).cpu()
        return take

    def finish(self):
        """Arguments of HeadTrainer.set_buffer (the de-duplicated layout of include/acez.h, acez_train_buffer)."""
        n_images = max(self._pose_inv) + 1
        pose_inv = torch.eye(4).repeat(n_images, 1, 1)
        for i, p in self._pose_inv.items():
            pose_inv[i] = p
        return {
            "features": self.features[:self.n], "target_px": self.target_px[:self.n], "view_idx": self.view_idx[:self.n],
            "view_aug_inv": torch.cat(self._aug), "view_K": torch.cat(self._K), "view_Kinv": torch.cat(self._Kinv),
            "view_image": torch.tensor(self._img, dtype=torch.int32), "image_pose_inv": pose_inv,
        }

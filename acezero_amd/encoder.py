"""Host side of the feature encoder: mirrors ace_network.Encoder (ace_network.py:14-59) over the C ABI (include/acez.h, section E).

    enc = Encoder.from_state_dict(torch.load("ace_encoder_pretrained.pt"))      # same keys as the reference's file
    feats = enc(image_B1HW)                  # == Encoder.forward: [B, 512, H/8, W/8]
    rows = enc.features_rows(image_B1HW)     # 16-bit [B*h*w, 512] rows (frame, y, x): training-buffer / head layout

dtype: "bf16" (default) or "fp16" -- the operand format the reference's autocast runs this network in (ace_trainer.py:366-367,
register_mapping.py:209-210); None reads $ACEZ_DTYPE. The rows must go to a head of the same dtype.

There is no CPU fallback: the HIP library does the work or the call raises.
"""
import ctypes as C

import torch

from . import _native as N

LAYER_NAMES = ["conv1", "conv2", "conv3", "conv4", "res1_conv1", "res1_conv2", "res1_conv3", "res2_conv1", "res2_conv2",
               "res2_conv3", "res2_skip"]


def output_size(h, w):
    oh, ow = C.c_int(0), C.c_int(0)
    N.check(N.lib().acez_encoder_output_size(int(h), int(w), C.byref(oh), C.byref(ow)))
    return oh.value, ow.value


class Encoder:
    OUTPUT_SUBSAMPLE = 8  # Regressor.OUTPUT_SUBSAMPLE, ace_network.py:159

    def __init__(self, state_dict, max_frames=16, max_h=480, max_w=640, device=None, dtype=None):
        from .head import DTYPES, resolve_dtype, torch_dtype
        self.dtype = resolve_dtype(dtype)
        self.feature_dtype = torch_dtype(self.dtype)
        if not torch.cuda.is_available():
            raise RuntimeError("acezero_amd.encoder needs a HIP device (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        ws, bs = [], []
        for name in LAYER_NAMES:
            w = state_dict[name + ".weight"].detach().to("cpu", torch.float32).contiguous()
            b = state_dict[name + ".bias"].detach().to("cpu", torch.float32).contiguous()
            ws.append(w)
            bs.append(b)
        self.out_channels = int(ws[9].shape[0])
        self._keep = (ws, bs)
        wp = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
        bp = (C.c_void_p * len(bs))(*[b.data_ptr() for b in bs])
        h = C.c_void_p()
        self.lib = N.lib()   # the library that owns the handle: pinned, like HeadTrainer.lib (N.diag_library() swaps the module-wide one)
        N.check(self.lib.acez_encoder_create(C.byref(h), wp, bp, self.out_channels, int(max_frames), int(max_h), int(max_w),
                                             DTYPES[self.dtype], self.device.index))
        self._h = h
        self.max_h, self.max_w, self.max_frames = int(max_h), int(max_w), int(max_frames)

    @classmethod
    def from_state_dict(cls, state_dict, **kw):
        """Accepts the encoder's own state_dict or a full Regressor state_dict ("encoder." prefix, ace_network.py:236-239)."""
        if any(k.startswith("encoder.") for k in state_dict):
            state_dict = {k[len("encoder."):]: v for k, v in state_dict.items() if k.startswith("encoder.")}
        return cls(state_dict, **kw)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.acez_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def features_rows(self, image_b1hw, out=None):
        """16-bit (self.feature_dtype) [B*h*w, C] rows in (frame, y, x) order; stays on the device."""
        assert image_b1hw.dim() == 4 and image_b1hw.shape[1] == 1, "expects grayscale [B,1,H,W]"
        img = image_b1hw.to(self.device, torch.float32).contiguous()
        b, _, h, w = img.shape
        oh, ow = output_size(h, w)
        if out is None:
            out = torch.empty((b * oh * ow, self.out_channels), dtype=self.feature_dtype, device=self.device)
        assert out.is_contiguous() and out.dtype == self.feature_dtype and out.shape == (b * oh * ow, self.out_channels)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        N.check(self.lib.acez_encoder_forward(self._h, img.data_ptr(), b, h, w, out.data_ptr(), C.c_void_p(stream)))
        return out

    def forward(self, image_b1hw):
        """Encoder.forward: [B, C, H/8, W/8] (fp32 container of the 16-bit values)."""
        b, _, h, w = image_b1hw.shape
        oh, ow = output_size(h, w)
        rows = self.features_rows(image_b1hw)
        return rows.view(b, oh, ow, self.out_channels).permute(0, 3, 1, 2).float()

    __call__ = forward

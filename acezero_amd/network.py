"""Host mirror of ace_network.Regressor (ace_network.py:150-270) for inference on one MI355X: HIP encoder + HIP head.

    net = Regressor.create_from_split_state_dict(encoder_state_dict, head_state_dict)
    sc = net(image_B1HW)                       # == Regressor.forward: float32 [B,3,H/8,W/8], stays on the device
    poses, inliers, masks = net.register(image_B1HW, intrinsics, ransac_params, seed)   # encoder -> head -> RANSAC in HBM

The reference moves the scene coordinates to the CPU and loops over frames (register_mapping.py:209-242); here a batch of
frames goes through three device-resident stages and only poses / inlier counts come back.
There is no CPU fallback.
"""
import ctypes as C
import re

import torch

from . import _native as N
from . import dsacstar
from .encoder import Encoder, output_size
from .head import HeadTrainer, _ptr, _stream, layer_names


class Regressor:
    OUTPUT_SUBSAMPLE = 8  # ace_network.py:159

    def __init__(self, encoder_state_dict, head_state_dict, max_frames=16, max_h=480, max_w=640, device=None, dtype=None):
        """dtype: "bf16" / "fp16" for encoder AND head (one format: the feature rows pass between them by raw pointer); None reads
        $ACEZ_DTYPE. fp16 is the reference's autocast arithmetic (register_mapping.py:209-210)."""
        hs = head_state_dict
        pattern = re.compile(r"^\d+c0\.weight$")                      # ace_network.py:207-208
        num_head_blocks = sum(1 for k in hs if pattern.match(k))
        use_homogeneous = hs["fc3.weight"].shape[0] == 4              # ace_network.py:211
        mean = hs["mean"].detach().float().view(3).cpu() if "mean" in hs else torch.zeros(3)
        kw = {}
        if use_homogeneous and "max_scale" in hs:
            kw = {"homogeneous_max_scale": float(hs["max_scale"]), "homogeneous_min_scale": float(hs["min_scale"])}
        oh, ow = output_size(max_h, max_w)
        self.encoder = Encoder.from_state_dict(encoder_state_dict, max_frames=max_frames, max_h=max_h, max_w=max_w, device=device, dtype=dtype)
        self.dtype = self.encoder.dtype
        self._enc_args = (max_frames, max_h, max_w)
        self.feature_dim = self.encoder.out_channels
        if self.feature_dim != 512:
            raise ValueError("the head kernels are built for 512 encoder features (ace_network.py:22 default)")
        # an inference-only context: max_batch rows per internal pass of the head (whole chunks of frames: with >= 32768 rows the
        # layers run on the encoder's large-tile kernels)
        # the head takes the encoder's operand format: the rows reach it by raw pointer (acez_head_forward_maps)
        self.heads = HeadTrainer(mean, num_head_blocks=num_head_blocks, use_homogeneous=use_homogeneous, max_batch=max(4, max_frames) * oh * ow,
                                 device=device, dtype=self.dtype, iterations=1, inference_only=True, **kw)
        self.heads.load_state_dict(hs)
        self.device = self.heads.device

    @classmethod
    def create_from_split_state_dict(cls, encoder_state_dict, head_state_dict, **kw):
        return cls(encoder_state_dict, head_state_dict, **kw)

    @classmethod
    def create_from_encoder(cls, encoder_state_dict, mean, num_head_blocks, use_homogeneous, seed=None, **kw):
        """ace_network.py:177-199: pre-trained encoder + a freshly initialised head (nn.Conv2d's default init: uniform in
        +-1/sqrt(fan_in) for weights and biases; `seed` makes it reproducible, the reference draws from the global generator)."""
        import math
        g = torch.Generator()
        if seed is not None:
            g.manual_seed(int(seed))
        no = 4 if use_homogeneous else 3
        bound = 1.0 / math.sqrt(512.0)
        hs = {}
        for name in layer_names(int(num_head_blocks)):
            hs[name + ".weight"] = (torch.rand((512, 512, 1, 1), generator=g) * 2 - 1) * bound
            hs[name + ".bias"] = (torch.rand((512,), generator=g) * 2 - 1) * bound
        hs["fc3.weight"] = (torch.rand((no, 512, 1, 1), generator=g) * 2 - 1) * bound
        hs["fc3.bias"] = (torch.rand((no,), generator=g) * 2 - 1) * bound
        hs["mean"] = torch.as_tensor(mean, dtype=torch.float32).view(1, 3, 1, 1).clone()
        return cls(encoder_state_dict, hs, **kw)

    def load_encoder(self, encoder_dict_file):
        """ace_network.py:253-257: replace the encoder weights from a checkpoint file."""
        sd = torch.load(encoder_dict_file, map_location="cpu")
        old = self.encoder
        self.encoder = Encoder.from_state_dict(sd, max_frames=self._enc_args[0], max_h=self._enc_args[1], max_w=self._enc_args[2],
                                               device=self.device.index, dtype=self.dtype)
        old.close()

    @classmethod
    def create_from_state_dict(cls, state_dict, **kw):
        enc = {k[len("encoder."):]: v for k, v in state_dict.items() if k.startswith("encoder.")}
        head = {k[len("heads."):]: v for k, v in state_dict.items() if k.startswith("heads.")}
        return cls(enc, head, **kw)

    def get_features(self, inputs):
        return self.encoder(inputs)

    def get_scene_coordinates(self, features_bchw):
        """Head.forward on a [B,512,h,w] feature tensor -> [B,3,h,w] (ace_network.py:262-263)."""
        b, c, h, w = features_bchw.shape
        rows = features_bchw.permute(0, 2, 3, 1).reshape(-1, c).to(self.device, self.encoder.feature_dtype).contiguous()
        return self._maps(rows, b, h, w)

    def _maps(self, rows, b, h, w):
        out = torch.empty((b, 3, h, w), dtype=torch.float32, device=self.device)
        N.check(self.heads.lib.acez_head_forward_maps(self.heads._h, _ptr(rows), int(b), int(h), int(w), _ptr(out), _stream()))
        return out

    def forward(self, inputs):
        b, _, h, w = inputs.shape
        oh, ow = output_size(h, w)
        rows = self.encoder.features_rows(inputs)
        return self._maps(rows, b, oh, ow)

    __call__ = forward

    def register(self, inputs, intrinsics, params, seed, frame_ids=None, want_masks=False, ransac_group=512):
        """images -> (poses [B,4,4] cam->world, inlier counts [B], masks): register_mapping.py:201-242 for a batch, all on device.

        Frames go through encoder + head in chunks of the context's max_frames; RANSAC is launched once per `ransac_group`
        frames (one 256-thread workgroup per frame: a launch wants hundreds of frames) on a side stream, so that the
        latency-bound RANSAC of one group overlaps the MFMA-bound encoder of the next."""
        b = int(inputs.shape[0])
        ids = list(range(b)) if frame_ids is None else list(frame_ids)
        main = torch.cuda.current_stream(self.device)
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream(self.device)
        poses, inl, masks, keep = [], [], [], []
        for g0 in range(0, b, ransac_group):
            g1 = min(b, g0 + ransac_group)
            sc = self.forward(inputs[g0:g1])
            ev = torch.cuda.Event()
            ev.record(main)
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                p, i, m = dsacstar.register_batch(sc, intrinsics[g0:g1], params, seed, ids[g0:g1], want_masks=want_masks)
            keep.append(sc)   # alive until the side stream has consumed it
            poses.append(p); inl.append(i); masks.append(m)
        main.wait_stream(self._side)
        for t in keep + poses + inl + [m for m in masks if m is not None]:
            t.record_stream(main)
        return torch.cat(poses), torch.cat(inl), (torch.cat(masks) if want_masks else None)

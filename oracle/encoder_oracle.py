"""CPU oracle of the ACE feature encoder -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(acezero_amd/) never does.  It restates ace_network.py:14-59 (Encoder: 11 convolutions, two residual blocks, stride 8)
with torch.nn.functional.conv2d on CPU.

Three arithmetic modes (same convention as oracle/head_oracle.py):
  * "fp32": no rounding anywhere.  PINNED against the reference itself: tests/golden/encoder_small.npz holds the output of
    the reference's own ace_network.Encoder (tests/golden/make_encoder_golden.py) for seeded weights and a seeded image;
    tests/test_encoder_oracle.py checks this oracle against it.
  * "bf16": weights, the input image and every stored activation are rounded to bfloat16 where the HIP kernels store
    them (NHWC bf16 activations, bf16 weight matrices); accumulation stays fp32.  This is what the GPU results are
    compared with.
  * "fp16": what the reference computes under torch.autocast(float16) (ace_trainer.py:366-367, register_mapping.py:209-210): image,
    weights AND biases are half tensors, every conv2d accumulates in fp32 and returns a half tensor, relu is exact, and the two residual
    sums add two half tensors (the activation is rounded before the add, the sum once more).  PINNED: tests/golden/encoder_small.npz
    also holds the reference Encoder's output under torch.autocast("cpu", dtype=torch.float16) (make_encoder_golden.py); the two agree
    to the last-place flips of oneDNN's accumulation order (tests/test_encoder_oracle.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

# (name, c_in, c_out, kernel, stride) in Encoder.__init__ order, ace_network.py:26-40
LAYERS = [
    ("conv1", 1, 32, 3, 1), ("conv2", 32, 64, 3, 2), ("conv3", 64, 128, 3, 2), ("conv4", 128, 256, 3, 2),
    ("res1_conv1", 256, 256, 3, 1), ("res1_conv2", 256, 256, 1, 1), ("res1_conv3", 256, 256, 3, 1),
    ("res2_conv1", 256, 512, 3, 1), ("res2_conv2", 512, 512, 1, 1), ("res2_conv3", 512, 512, 3, 1),
    ("res2_skip", 256, 512, 1, 1),
]


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def fp16_round(x):
    return x.to(torch.float16).to(torch.float32)


def init_weights(seed=4099, out_channels=512):
    """Seeded state_dict with the reference's key names (generator: acezero_amd.synth.init_encoder_weights)."""
    from acezero_amd import synth
    return {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights(seed, out_channels).items()}


class EncoderOracle:
    def __init__(self, state_dict, mode="fp32"):
        assert mode in ("fp32", "bf16", "fp16")
        self.mode = mode
        self.sd = {k: v.detach().to(torch.float32).clone() for k, v in state_dict.items()}

    def _r(self, x):
        return bf16_round(x) if self.mode == "bf16" else (fp16_round(x) if self.mode == "fp16" else x)

    def _conv(self, x, name, stride, pad):
        w = self._r(self.sd[name + ".weight"])
        b = self.sd[name + ".bias"]
        return F.conv2d(x, w, fp16_round(b) if self.mode == "fp16" else b, stride=stride, padding=pad)

    def forward(self, image_b1hw):
        """ace_network.py:42-59.  Returns features [B, C, H/8, W/8] (fp32 container of the 16-bit values in bf16 / fp16 mode)."""
        r = self._r
        x = r(image_b1hw.to(torch.float32))
        x = r(F.relu(self._conv(x, "conv1", 1, 1)))
        x = r(F.relu(self._conv(x, "conv2", 2, 1)))
        x = r(F.relu(self._conv(x, "conv3", 2, 1)))
        res = r(F.relu(self._conv(x, "conv4", 2, 1)))
        x = r(F.relu(self._conv(res, "res1_conv1", 1, 1)))
        x = r(F.relu(self._conv(x, "res1_conv2", 1, 0)))
        # bf16: the kernel adds the residual in fp32 before the single bf16 store of `res`; fp16: relu(conv) is a half tensor before the add
        pre = r if self.mode == "fp16" else (lambda t: t)
        res = r(res + pre(F.relu(self._conv(x, "res1_conv3", 1, 1))))
        x = r(F.relu(self._conv(res, "res2_conv1", 1, 1)))
        x = r(F.relu(self._conv(x, "res2_conv2", 1, 0)))
        skip = r(self._conv(res, "res2_skip", 1, 0))
        x = r(skip + pre(F.relu(self._conv(x, "res2_conv3", 1, 1))))
        return x

    def features_rows(self, image_b1hw):
        """[B*h*w, C] rows in pixel order (frame, y, x): the layout of the training buffer and of acez_head_forward."""
        f = self.forward(image_b1hw)
        return f.permute(0, 2, 3, 1).reshape(-1, f.shape[1]).contiguous()

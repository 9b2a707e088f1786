"""Generate tests/golden/cloud_cases.npz by running the REFERENCE's own point-cloud extraction on CPU.

Run in the build container only (needs /root/reference):   python tests/golden/make_cloud_golden.py

What runs is the unmodified ace_vis_util.get_point_cloud_from_network (ace_vis_util.py:430-591) with a stand-in network
(returns a prepared scene-coordinate map per frame) and a stand-in data loader (len() = the mapping-sequence length that
sets the per-image point budgets, iteration = the few frames of the case).  Modules the container lacks are stubbed;
skimage's imread / resize are replaced by functions returning arrays of the right shape (colours are not part of the
fixture).  Inputs come from acezero_amd.synth with the seeds below, so the fixture only holds the reference's OUTPUT
point clouds.
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"

for name in ["trimesh", "pyrender", "matplotlib.pyplot"]:
    sys.modules.setdefault(name, MagicMock())
_sk = types.ModuleType("skimage")
_io = types.ModuleType("skimage.io")
_color = types.ModuleType("skimage.color")
_tf = types.ModuleType("skimage.transform")
_io.imread = lambda path: np.zeros(tuple(int(v) for v in os.path.basename(path).split("x")) + (3,), np.uint8)
_color.gray2rgb = lambda a: np.stack([a] * 3, -1)


def _resize(a, shape):
    ys = (np.arange(shape[0]) * a.shape[0] // shape[0])
    xs = (np.arange(shape[1]) * a.shape[1] // shape[1])
    return a[ys][:, xs]


_tf.resize = _resize
_sk.io, _sk.color, _sk.transform = _io, _color, _tf
sys.modules.update({"skimage": _sk, "skimage.io": _io, "skimage.color": _color, "skimage.transform": _tf})
sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self   # ace_vis_util.py:472-474 move the inputs to the GPU

import ace_vis_util  # noqa: E402

from tests.helpers import CLOUD_CASES as CASES, cloud_case_inputs as case_inputs  # noqa: E402


class _Net:
    OUTPUT_SUBSAMPLE = 8

    def __init__(self, maps):
        self.maps, self.i = maps, 0

    def __call__(self, image):
        out = torch.from_numpy(self.maps[self.i:self.i + 1].copy())
        self.i += 1
        return out


class _Loader:
    def __init__(self, maps, poses_inv, Ks, length):
        self.maps, self.poses_inv, self.Ks, self.length = maps, poses_inv, Ks, length

    def __len__(self):
        return self.length

    def __iter__(self):
        n, _, h, w = self.maps.shape
        for i in range(n):
            image = torch.zeros(1, 1, h * 8, w * 8)
            yield (image, None, torch.from_numpy(self.poses_inv[i:i + 1]), None, torch.from_numpy(self.Ks[i:i + 1]), None, None,
                   ["%dx%d" % (h * 8, w * 8)], None)


def run_reference(name):
    maps, poses_inv, Ks, loader_len, depth, dense = case_inputs(name)
    torch.manual_seed(1)
    xyz, clr = ace_vis_util.get_point_cloud_from_network(_Net(maps), _Loader(maps, poses_inv, Ks, loader_len), depth, dense)
    assert clr.shape == xyz.shape
    return np.ascontiguousarray(xyz, dtype=np.float32)


if __name__ == "__main__":
    out = {}
    for name in CASES:
        out[name] = run_reference(name)
        print(name, out[name].shape)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cloud_cases.npz"), **out)

"""Point-cloud extraction on the GPU through the C ABI: bit-exact against oracle/cloud_oracle.py and
against the reference function's own output (tests/golden/cloud_cases.npz)."""
import os

import numpy as np
import pytest
import torch

from acezero_amd import synth
from oracle import cloud_oracle as co
from tests import helpers

pytestmark = pytest.mark.gpu


def _gpu(sc, pinv, K, depth, dense, loader_len, seed=5, first=0, opengl=True):
    from acezero_amd import pointcloud as pc
    xyz, src, counts, keep = pc.filter_scene_coordinates(torch.from_numpy(np.ascontiguousarray(sc)).cuda(), torch.from_numpy(pinv).cuda(),
                                                         torch.from_numpy(np.ascontiguousarray(K)).cuda(), depth, dense, loader_len, seed=seed,
                                                         first_frame_id=first, opengl=opengl)
    torch.cuda.synchronize()
    return xyz.cpu().numpy(), src.cpu().numpy(), counts.cpu().numpy(), keep.cpu().numpy().astype(bool)


def _same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("name", list(helpers.CLOUD_CASES))
def test_golden_cases_bit_exact(golden_dir, name):
    sc, pinv, K, loader_len, depth, dense = helpers.cloud_case_inputs(name)
    xyz, src, counts, keep = _gpu(sc, pinv, K, depth, dense, loader_len)
    oxyz, osrc, ocounts, okeep = co.point_cloud(sc, pinv[:, :3], K, depth, dense, loader_len, 5, 0)
    assert np.array_equal(keep, okeep) and np.array_equal(counts, ocounts) and np.array_equal(src, osrc) and _same(xyz, oxyz)
    ref = np.load(os.path.join(golden_dir, "cloud_cases.npz"))[name]
    if name in helpers.CLOUD_RANDOM_CASES:
        assert len(xyz) == len(ref)
    else:
        assert _same(xyz, ref)          # the reference's own point cloud


def test_full_size_batch_all_branches_and_conventions():
    n = 96
    fr = synth.make_registration_frames(seed=77, n_frames=n, h=60, w=80, noise_sigma=0.0015, outlier_ratio=0.25)
    # make the frames exercise different branches: noisier ones (relaxed), cleaner ones (sub-sampled at 1000 frames -> max 1000)
    rng = np.random.default_rng(3)
    sc = fr["scene_coords"].copy()
    sc[::3] += rng.normal(0, 0.03, size=sc[::3].shape).astype(np.float32)
    pinv = np.linalg.inv(fr["poses"]).astype(np.float32)
    K = np.stack([np.array([[fr["focal"], 0, fr["ppx"]], [0, fr["focal"], fr["ppy"]], [0, 0, 1]], np.float32)] * n)
    branches = set()
    for loader_len, depth in ((1000, 100.0), (100, 3.0)):
        pmin, pmax = co.points_per_image(loader_len)
        for f in range(0, n, 7):
            branches.add(co.filter_frame(sc[f], pinv[f, :3], K[f], depth, False, pmin, pmax, 11, 40 + f)[1]["branch"])
        xyz, src, counts, keep = _gpu(sc, pinv, K, depth, False, loader_len, seed=11, first=40)
        oxyz, osrc, ocounts, okeep = co.point_cloud(sc, pinv[:, :3], K, depth, False, loader_len, 11, 40)
        assert np.array_equal(keep, okeep) and np.array_equal(counts, ocounts) and np.array_equal(src, osrc) and _same(xyz, oxyz)
    assert branches == {"plain", "relaxed", "subsampled"}
    xyz_cv, *_ = _gpu(sc, pinv, K, 3.0, False, 100, seed=11, first=40, opengl=False)
    assert _same(xyz_cv * np.array([1, -1, -1], np.float32), xyz)
    # size-independent properties: source indices strictly increasing, every point is the coordinate of its source pixel
    assert np.all(np.diff(src) > 0)
    flat = sc.transpose(0, 2, 3, 1).reshape(-1, 3)
    assert _same(flat[src], xyz_cv)


def test_nan_inf_degenerate_and_largest_map():
    h, w = 128, 192                                   # 24576 pixels: the largest map a workgroup's LDS holds
    fr = synth.make_registration_frames(seed=5, n_frames=2, h=h, w=w, noise_sigma=0.001, outlier_ratio=0.1)
    sc = fr["scene_coords"].copy()
    sc[0, 0, 10, 10] = np.nan
    sc[0, 1, 50, 60] = np.inf
    sc[1, :, :, :] = 0.0                              # a constant map: zero gradient, every pixel projects to the same point
    pinv = np.linalg.inv(fr["poses"]).astype(np.float32)
    K = np.stack([np.array([[fr["focal"], 0, fr["ppx"]], [0, fr["focal"], fr["ppy"]], [0, 0, 1]], np.float32)] * 2)
    for loader_len in (10, 400):
        xyz, src, counts, keep = _gpu(sc, pinv, K, 100.0, False, loader_len)
        oxyz, osrc, ocounts, okeep = co.point_cloud(sc, pinv[:, :3], K, 100.0, False, loader_len, 5, 0)
        assert np.array_equal(keep, okeep) and np.array_equal(counts, ocounts) and np.array_equal(src, osrc) and _same(xyz, oxyz)
    assert not keep[0, 10 * w + 10] and not keep[0, 50 * w + 60]


def test_argument_validation():
    from acezero_amd import _native as N
    from acezero_amd import pointcloud as pc
    sc = torch.zeros(1, 3, 2, 8, device="cuda")
    with pytest.raises(N.AcezError):
        pc.filter_scene_coordinates(sc, torch.eye(4).unsqueeze(0), torch.eye(3).unsqueeze(0), 100.0, False, 10)
    with pytest.raises(N.AcezError):
        pc.filter_scene_coordinates(torch.zeros(1, 3, 160, 160, device="cuda"), torch.eye(4).unsqueeze(0), torch.eye(3).unsqueeze(0), 100.0, False, 10)
    with pytest.raises(RuntimeError):
        pc.filter_scene_coordinates(torch.zeros(1, 3, 8, 8), torch.eye(4).unsqueeze(0), torch.eye(3).unsqueeze(0), 100.0, False, 10)


def test_network_to_point_cloud_matches_oracle_on_the_device_maps():
    from acezero_amd import pointcloud as pc
    from acezero_amd.network import Regressor
    from oracle import encoder_oracle
    esd = encoder_oracle.init_weights(seed=4099)
    hflat = synth.init_head_params(3)
    hsd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict(hflat, mean=(1.0, -2.0, 0.5)).items()}
    net = Regressor.create_from_split_state_dict(esd, hsd, max_frames=4, max_h=64, max_w=96)
    imgs = torch.from_numpy(synth.make_gray_images(seed=21, n=6, h=64, w=96))
    cams = synth.random_cameras(np.random.default_rng(2), 6)
    pinv = torch.from_numpy(np.linalg.inv(cams).astype(np.float32))
    K = torch.tensor([[80.0, 0, 48], [0, 80.0, 32], [0, 0, 1]]).repeat(6, 1, 1)
    rgb = np.random.default_rng(4).integers(0, 256, size=(6, 64, 96, 3), dtype=np.uint8)
    frames = [(imgs[i:i + 3], pinv[i:i + 3], K[i:i + 3], rgb[i:i + 3]) for i in (0, 3)]
    xyz, clr = pc.get_point_cloud_from_network(net, frames, filter_depth=100, dense_cloud=True, seed=7)
    sc = net(imgs[:3].cuda()).cpu().numpy(), net(imgs[3:].cuda()).cpu().numpy()
    parts, cols = [], []
    for b, first in ((0, 0), (1, 3)):
        oxyz, osrc, _, _ = co.point_cloud(sc[b], pinv[first:first + 3, :3].numpy(), K[first:first + 3].numpy(), 100, True, 6, 7, first)
        parts.append(oxyz)
        f, p = np.divmod(osrc, 8 * 12)
        y, x = np.divmod(p, 12)
        cols.append(rgb[first + f, y * 8 + 4, x * 8 + 4].astype(np.float64))
    assert _same(xyz, np.concatenate(parts)) and np.array_equal(clr, np.concatenate(cols))
    assert len(xyz) > 0 and clr.shape == xyz.shape

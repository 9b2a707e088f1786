"""CPU: the point-cloud filter oracle (oracle/cloud_oracle.py) pinned on the reference function's own output
(tests/golden/cloud_cases.npz, made by tests/golden/make_cloud_golden.py from ace_vis_util.get_point_cloud_from_network),
plus the laws of the counter-based sub-sampling and the host writers."""
import os

import numpy as np
import pytest

from oracle import cloud_oracle as co
from tests import helpers


def _run(name, seed=5):
    sc, pinv, K, loader_len, depth, dense = helpers.cloud_case_inputs(name)
    return sc, co.point_cloud(sc, pinv[:, :3], K, depth, dense, loader_len, seed, 0)


@pytest.mark.parametrize("name", [n for n in helpers.CLOUD_CASES if n not in helpers.CLOUD_RANDOM_CASES])
def test_oracle_reproduces_reference_point_cloud_exactly(golden_dir, name):
    ref = np.load(os.path.join(golden_dir, "cloud_cases.npz"))[name]
    _, (xyz, src, counts, keep) = _run(name)
    assert xyz.shape == ref.shape and np.array_equal(xyz, ref)     # copies of the input coordinates: equality == same keep masks
    assert counts.sum() == len(xyz) == len(src) == keep.sum()


@pytest.mark.parametrize("name", helpers.CLOUD_RANDOM_CASES)
def test_random_branch_same_count_and_candidates_as_reference(golden_dir, name):
    ref = np.load(os.path.join(golden_dir, "cloud_cases.npz"))[name]
    sc, (xyz, src, counts, keep) = _run(name)
    assert len(xyz) == len(ref)                                    # k = int(max / n_valid * n_valid) per frame
    _, pinv, K, loader_len, depth, dense = helpers.cloud_case_inputs(name)
    pmin, _ = co.points_per_image(loader_len)
    # both subsets are drawn from the same candidate set: rerun with an unlimited budget and check membership
    cand = set()
    for f in range(len(sc)):
        k, info = co.filter_frame(sc[f], pinv[f, :3], K[f], depth, dense, pmin, 10 ** 9)
        assert info["branch"] == "plain"
        pts = sc[f].reshape(3, -1)[:, k].T * np.array([1, -1, -1], np.float32)
        cand.update(map(bytes, pts.astype(np.float32)))
    assert all(bytes(p) in cand for p in ref.astype(np.float32))
    assert all(bytes(p) in cand for p in xyz.astype(np.float32))


def test_subsampling_is_a_uniform_k_subset_keyed_by_frame_and_seed():
    sc, pinv, K, loader_len, depth, dense = helpers.cloud_case_inputs("subsampled")
    pmin, pmax = co.points_per_image(loader_len)
    hits = np.zeros(sc.shape[2] * sc.shape[3])
    trials = 300
    for fid in range(trials):
        keep, info = co.filter_frame(sc[0], pinv[0, :3], K[0], depth, dense, pmin, pmax, seed=9, frame_id=fid)
        assert info["branch"] == "subsampled" and keep.sum() == info["k"] == pmax
        hits += keep
    full, _ = co.filter_frame(sc[0], pinv[0, :3], K[0], depth, dense, pmin, 10 ** 9)
    assert hits[~full].sum() == 0
    p = pmax / full.sum()
    freq = hits[full] / trials
    assert abs(freq.mean() - p) < 1e-9 and freq.std() < 2.5 * np.sqrt(p * (1 - p) / trials)
    a, _ = co.filter_frame(sc[0], pinv[0, :3], K[0], depth, dense, pmin, pmax, seed=9, frame_id=1)
    b, _ = co.filter_frame(sc[0], pinv[0, :3], K[0], depth, dense, pmin, pmax, seed=10, frame_id=1)
    assert not np.array_equal(a, b)


def test_gradient_uses_reflect_padding_and_nan_never_passes():
    sc, pinv, K, *_ = helpers.cloud_case_inputs("plain")
    m = sc[0].copy()
    err, grad, z = co.frame_quantities(m, pinv[0, :3], K[0])
    h, w = m.shape[1:]
    g = grad.reshape(h, w)
    d = np.linalg.norm(m[:, 3, 2] - m[:, 3, 1])
    assert np.isclose(g[3, 0], max(d, np.linalg.norm(m[:, 3, 0] - m[:, 2, 0])), rtol=1e-6)
    m[0, 5, 5] = np.nan
    keep, _ = co.filter_frame(m, pinv[0, :3], K[0], 100.0, False, 100, 1000)
    assert not keep[5 * w + 5] and not keep[5 * w + 6] and not keep[6 * w + 5]


def test_writers_txt_and_ply(tmp_path):
    from acezero_amd import pointcloud as pc
    xyz = np.array([[0.5, -1.25, 3.0], [1e-3, 2.0, -7.5]], np.float32)
    clr = np.array([[0.4, 127.6, 255.0], [12.0, 13.0, 14.0]])
    pc.write_point_cloud(tmp_path / "a.txt", xyz, clr)
    lines = open(tmp_path / "a.txt").read().splitlines()
    assert lines[0] == "0.5 -1.25 3.0 0 128 255" and lines[1].endswith("12 13 14")      # export_point_cloud.py:113-114
    pc.write_point_cloud(tmp_path / "a.ply", xyz, clr)
    raw = open(tmp_path / "a.ply", "rb").read()
    head, body = raw.split(b"end_header\n")
    assert b"element vertex 2" in head and b"binary_little_endian" in head and len(body) == 2 * 16
    assert np.array_equal(np.frombuffer(body[:12], "<f4"), xyz[0]) and tuple(body[12:16]) == (0, 128, 255, 255)
    with pytest.raises(ValueError):
        pc.write_point_cloud(tmp_path / "a.obj", xyz, clr)
    assert pc.points_per_image(1000) == co.points_per_image(1000) == (100, 1000)

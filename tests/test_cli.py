"""CPU: the command-line surfaces keep every flag, default and choice of the reference (golden captured from the
reference's argparse by tests/golden/make_cli_golden.py); pose-file lines round-trip through the reference format.
GPU: a small mapping + registration run end to end through the two scripts."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from acezero_amd import cli, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _surface(parser):
    out = {}
    for act in parser._actions:
        if act.dest == "help":
            continue
        d = act.default
        if isinstance(d, Path):
            d = str(d)
        out[act.dest] = {"flags": list(act.option_strings), "default": d, "choices": list(act.choices) if act.choices else None,
                         "positional": not act.option_strings}
    return out


@pytest.mark.parametrize("name,parser,extra", [("train_ace", cli.train_parser, {"feature_buffer", "num_gpus", "compute_dtype"}),
                                               ("register_mapping", cli.register_parser, {"feature_file", "compute_dtype"}),
                                               ("ace_zero", cli.ace_zero_parser, {"encoder_path", "compute_dtype"}),
                                               ("export_point_cloud", cli.export_point_cloud_parser, {"compute_dtype"})])
def test_flag_surface_matches_reference(name, parser, extra, golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "cli_flags.json")))[name]
    mine = _surface(parser())
    assert set(mine) - set(ref) == extra           # additive flags only
    for dest, spec in ref.items():
        assert dest in mine, dest
        m = mine[dest]
        assert m["flags"] == spec["flags"] and m["positional"] == spec["positional"], dest
        assert m["choices"] == spec["choices"], dest
        assert m["default"] == spec["default"], (dest, m["default"], spec["default"])


def test_pose_line_format_roundtrip(tmp_path):
    from scipy.spatial.transform import Rotation
    R = Rotation.from_euler("xyz", [0.3, -1.1, 2.0]).as_matrix()
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = [1.5, -2.0, 0.25]
    p = tmp_path / "poses.txt"
    with open(p, "w") as f:
        cli.write_pose_line(f, "a/b.png", T, 1234, 525.0)
    tok = open(p).read().split()
    assert len(tok) == 10 and tok[0] == "a/b.png"                      # dataset_io.py:128 asserts 10 tokens
    qw, qx, qy, qz = [float(x) for x in tok[1:5]]
    np.testing.assert_allclose(Rotation.from_quat([qx, qy, qz, qw]).as_matrix(), R, atol=1e-12)
    np.testing.assert_allclose([float(x) for x in tok[5:8]], T[:3, 3])
    assert float(tok[8]) == 525.0 and float(tok[9]) == 1234


def test_missing_feature_file_is_a_clear_error():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_ace.py"), "x/*.png", "/tmp/out.pt"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "no files match" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_train_then_register_end_to_end(tmp_path):
    import torch
    prob = synth.make_training_problem(seed=4, n_images=8, views_per_image=2, patches_per_view=512)
    buf = tmp_path / "buffer.npz"
    cli.save_feature_buffer(buf, prob)
    out = tmp_path / "scene.pt"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_ace.py"), "synthetic/*.png", str(out), "--feature_buffer", str(buf),
                        "--iterations", "60", "--iterations_output", "20", "--learning_rate_schedule", "1cyclepoly", "--learning_rate_max", "0.003",
                        "--learning_rate_warmup_iterations", "10", "--learning_rate_cooldown_iterations", "20", "--repro_loss_type", "tanh",
                        "--batch_size", "1024", "--pose_refinement", "mlp", "--refine_calibration", "True", "--use_external_focal_length", "525"],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    sd = torch.load(out, map_location="cpu")
    assert sd["res3_conv1.weight"].dtype == torch.float16 and sd["fc3.weight"].shape == (4, 512, 1, 1) and "mean" in sd
    lines = open(tmp_path / "poses_scene_preliminary.txt").read().strip().split("\n")
    assert len(lines) == 8 and all(len(l.split()) == 10 and l.split()[-1] == "inf" for l in lines)
    rows = [l.split() for l in open(tmp_path / "scene.txt").read().strip().split("\n")]
    assert len(rows) == 3 and len(rows[0]) == 8                         # iter time loss inliers pose_mean pose_min pose_max focal
    # registration of synthetic scene-coordinate maps through the second script
    fr = synth.make_registration_frames(seed=6, n_frames=5)
    ff = tmp_path / "frames.npz"
    np.savez(ff, scene_coordinates=fr["scene_coords"], focal=np.float32(fr["focal"]), ppx=np.float32(fr["ppx"]), ppy=np.float32(fr["ppy"]),
             image_files=np.array([f"f{i}.png" for i in range(5)]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "register_mapping.py"), "synthetic/*.png", str(out), "--feature_file", str(ff),
                        "--session", "test", "--hypotheses", "32", "--hypotheses_max_tries", "16"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l.split() for l in open(tmp_path / "poses_test.txt").read().strip().split("\n")]
    assert len(lines) == 5
    from scipy.spatial.transform import Rotation
    for i, t in enumerate(lines):
        T = np.eye(4)
        T[:3, :3] = Rotation.from_quat([float(t[2]), float(t[3]), float(t[4]), float(t[1])]).as_matrix()
        T[:3, 3] = [float(x) for x in t[5:8]]
        np.testing.assert_allclose(np.linalg.inv(T)[:3, 3], fr["poses"][i][:3, 3], atol=0.03)   # file stores world->cam
        assert float(t[9]) > 1000
    # --max_estimates: a seeded random subset in file order (register_mapping.py:122-147,256), the same frames get the same poses
    r = subprocess.run([sys.executable, os.path.join(ROOT, "register_mapping.py"), "synthetic/*.png", str(out), "--feature_file", str(ff),
                        "--session", "sub", "--hypotheses", "32", "--hypotheses_max_tries", "16", "--max_estimates", "3"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    sub = [l.split() for l in open(tmp_path / "poses_sub.txt").read().strip().split("\n")]
    want = sorted(torch.randperm(5, generator=torch.Generator().manual_seed(1305))[:3].tolist())   # register_mapping.py default --base_seed
    assert [t[0] for t in sub] == [f"f{i}.png" for i in want]
    full = {t[0]: t for t in lines}
    assert all(t == full[t[0]] for t in sub)


def test_pose_files_are_the_reference_format_both_ways(golden_dir, tmp_path):
    """tests/golden/pose_file_ref.txt was written by the reference's write_pose_to_pose_file and parsed by its load_dataset_ace
    (make_pose_file_golden.py): our writer must produce the same bytes, our reader the same entries."""
    from tests.helpers import pose_file_cases
    P, conf = pose_file_cases()
    out = tmp_path / "mine.txt"
    with open(out, "w") as f:
        for i in range(len(P)):
            cli.write_pose_line(f, f"scene/frame_{i:03d}.png", P[i], int(conf[i]) if i else float("inf"), 525.0 + i)
    assert open(out).read() == open(os.path.join(golden_dir, "pose_file_ref.txt")).read()
    ref = np.load(os.path.join(golden_dir, "pose_file_ref.npz"))
    files, c2w, focals = cli.read_ace_pose_file(os.path.join(golden_dir, "pose_file_ref.txt"), 500)
    assert files == [str(x) for x in ref["files"]] and np.allclose(focals, ref["focals"])
    assert np.allclose(c2w, ref["c2w"], atol=1e-6)          # the reference returns float32 matrices
    assert "scene/frame_001.png" not in files and "scene/frame_000.png" in files and "scene/frame_002.png" in files   # 499 dropped; inf, 500 kept


def test_export_point_cloud_from_visualization_buffer(tmp_path):
    """The host-only branch of export_point_cloud.py (:95-107): a pickled {'map_xyz', 'map_clr'} buffer in OpenGL coordinates."""
    import pickle
    xyz = np.array([[1.0, 2.0, 3.0], [-0.5, 0.25, 4.0]], np.float32)
    clr = np.array([[10.0, 20.0, 30.0], [200.0, 100.0, 0.0]])
    with open(tmp_path / "buf.pkl", "wb") as f:
        pickle.dump({"map_xyz": xyz, "map_clr": clr}, f)
    assert cli.export_point_cloud_main([str(tmp_path / "pc.txt"), "--visualization_buffer", str(tmp_path / "buf.pkl"), "--convention", "opencv"]) == 0
    rows = [line.split() for line in open(tmp_path / "pc.txt").read().splitlines()]
    assert [float(v) for v in rows[0][:3]] == [1.0, -2.0, -3.0] and rows[1][3:] == ["200", "100", "0"]
    with pytest.raises(SystemExit):
        cli.export_point_cloud_main([str(tmp_path / "pc.txt")])                                    # neither buffer nor network + pose file
    with pytest.raises(SystemExit):
        cli.export_point_cloud_main([str(tmp_path / "pc.txt"), "--visualization_buffer", str(tmp_path / "buf.pkl"), "--dense_point_cloud", "True"])
    c = cli.source_colours(np.arange(2 * 16 * 24 * 3, dtype=np.uint8).reshape(2, 16, 24, 3), np.array([1, 0]), np.array([4, 0]), 3)
    assert c.shape == (2, 3) and np.array_equal(c[1], np.arange(2 * 16 * 24 * 3, dtype=np.uint8).reshape(2, 16, 24, 3)[0, 4, 4])


def test_reference_module_names_are_importable():
    """register_mapping.py:12 does `import dsacstar`, train_ace.py:20 `from ace_trainer import TrainerACE` (surface names)."""
    import inspect
    import dsacstar
    from ace_trainer import TrainerACE
    sig = inspect.signature(dsacstar.forward_rgb)
    assert list(sig.parameters) == ["sceneCoordinates", "outPose", "ransacHypotheses", "inlierThreshold", "focalLength", "ppointX", "ppointY",
                                    "inlierAlpha", "maxReproj", "subSampling", "randomSeed", "max_hypotheses_tries"]   # dsacstar.cpp:66-78
    from acezero_amd import cli
    opt = cli.train_parser().parse_args(["frames/*.png", "out/map.pt"])
    tr = TrainerACE(opt)
    assert hasattr(tr, "train") and tr.options is opt
    opt.batch_size = 5000
    with pytest.raises(ValueError):
        TrainerACE(opt)


def test_use_half_false_is_refused_not_silently_ignored(tmp_path):
    """train_ace.py --use_half False selects fp32 arithmetic in the reference (ace_trainer.py:330); this package has no fp32 head path
    and must say so instead of running in 16 bits (missing 2)."""
    opt = cli.train_parser().parse_args(["scene/*.png", str(tmp_path / "out.pt"), "--use_half", "False"])
    with pytest.raises(SystemExit) as e:
        cli.train_with_options(opt)
    assert "not implemented" in str(e.value) and "use_half" in str(e.value)


def test_frame_size_classes_group_the_sorted_file_list(tmp_path):
    """register_mapping.py on a folder of mixed frame sizes: one class per image size, positions in the sorted list (headers only)."""
    from PIL import Image
    from acezero_amd import cli
    sizes = [(64, 48), (80, 48), (64, 48), (48, 64), (80, 48)]
    for i, wh in enumerate(sizes):
        Image.new("RGB", wh).save(tmp_path / f"im_{i:02d}.png")
    files, classes = cli.frame_size_classes(str(tmp_path / "im_*.png"))
    assert [os.path.basename(f) for f in files] == [f"im_{i:02d}.png" for i in range(5)]
    assert classes == {(64, 48): [0, 2], (80, 48): [1, 4], (48, 64): [3]}
    with pytest.raises(SystemExit):
        cli.frame_size_classes(str(tmp_path / "nothing_*.png"))


def test_gauge_aware_pose_comparison_is_invariant_to_a_similarity():
    """tools/pose_geometry.py (the session studies of DESIGN.md section 4c): a similarity transform of the whole reconstruction gives
    zero error and the inverse scale; a distortion of the centres shows up in the centre error, not in the relative rotations."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.pose_geometry import geometry
    rng = np.random.default_rng(5)

    def rot(r):
        th = np.linalg.norm(r); k = r / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    gt = np.tile(np.eye(4), (40, 1, 1))
    for i in range(40):
        gt[i, :3, :3] = rot(rng.normal(size=3)); gt[i, :3, 3] = rng.normal(size=3)
    Q, s, t = rot(np.array([0.3, -0.2, 0.5])), 2.5, np.array([1.0, -2.0, 3.0])
    est = gt.copy()
    est[:, :3, :3] = Q @ gt[:, :3, :3]
    est[:, :3, 3] = s * (gt[:, :3, 3] @ Q.T) + t
    g = geometry(est, gt)
    assert abs(g["scale"] - 1 / s) < 1e-3 and g["centre_rel_median"] < 1e-6 and g["rot_abs_deg_median"] < 1e-3 and g["rot_rel_deg_median"] < 1e-3
    est[:, :3, 3] += rng.normal(size=(40, 3)) * 0.5
    g = geometry(est, gt)
    assert g["centre_rel_median"] > 0.01 and g["rot_rel_deg_median"] < 1e-3

"""End-to-end on the GPU (SURVEY 8f N1-N4 together): a view-consistent synthetic room, the stand-in encoder, then
(1) mapping with known poses + relocalisation of held-out frames, (2) the whole ACE0 loop from a depth-supervised seed in one
process. These are functional tests of the product (accuracy against ground-truth cameras), not parity tests."""
import os

import numpy as np
import pytest
import torch

from acezero_amd import synth

pytestmark = pytest.mark.gpu


def _pose_err(est, gt):
    dt = np.linalg.norm(est[:, :3, 3] - gt[:, :3, 3], axis=1)
    R = np.einsum("nij,nkj->nik", est[:, :3, :3], gt[:, :3, :3])
    ang = np.degrees(np.arccos(np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1, 1)))
    return dt, ang


def _align_similarity(est, gt):
    """Least-squares similarity (Umeyama) of the camera centres, applied to the estimated poses (as eval_poses.py aligns before
    it measures)."""
    a, b = est[:, :3, 3], gt[:, :3, 3]
    ma, mb = a.mean(0), b.mean(0)
    H = (b - mb).T @ (a - ma) / len(a)
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(U) * np.linalg.det(Vt))])
    R = U @ D @ Vt
    s = np.trace(np.diag(S) @ D) / ((a - ma) ** 2).sum(1).mean()
    out = est.copy()
    out[:, :3, :3] = R @ est[:, :3, :3]
    out[:, :3, 3] = (s * (R @ (a - ma).T)).T + mb
    return out, s


@pytest.fixture(scope="module")
def room():
    seq = synth.render_room_sequence(seed=2089, n_frames=72, arc_deg=36.0, device="cuda")
    esd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights_bandpass(seed=4099).items()}
    return seq, esd


def _opt(seq, **kw):
    from acezero_amd.session import default_options
    it = 3000
    base = dict(use_external_focal_length=seq["focal"], try_seeds=2, seed_iterations=it, iterations=it, refit_iterations=it, iterations_max=8,
                final_refit_posewait=it // 5, learning_rate_warmup_iterations=it // 5, cooldown_iterations=it // 5, aug_rotation=2,
                aug_scale=1.06, aug_black_white=0.02)
    base.update(kw)
    return default_options(**base)


def test_mapping_with_known_poses_relocalises_held_out_frames(room):
    from acezero_amd.session import ReconstructionSession
    seq, esd = room
    n = seq["images"].shape[0]
    ses = ReconstructionSession(esd, seq["images"], opt=_opt(seq), depth=seq["depth"])
    even = list(range(0, n, 2))
    m = ses.map(even, seq["poses"][even].cpu(), seq["focal"], iterations=3000, loss_type="tanh", schedule="1cyclepoly", lr_max=0.003)
    assert m["iterations"] <= 3000 and m["batch_inliers"] > 0.5 and m["buffer"] == len(even) * 10 * 1024
    poses, inl = ses.register(m["head"], seq["focal"])
    dt, ang = _pose_err(poses, seq["poses"].cpu().numpy())
    odd = np.arange(1, n, 2)
    assert (inl > 500).mean() >= 0.97
    assert np.median(dt[odd]) < 0.01 and np.median(ang[odd]) < 0.5, (np.median(dt[odd]), np.median(ang[odd]))     # < 1 cm, < 0.5 deg
    assert np.median(dt[even]) < 0.01 and np.median(ang[even]) < 0.5
    # the refined-pose output of a run without refinement is the input (world -> camera)
    w2c = np.linalg.inv(seq["poses"][even].cpu().numpy().astype(np.float64))[:, :3]
    assert np.allclose(m["poses_w2c"], w2c, atol=1e-5)


def test_ace_zero_loop_reconstructs_the_sequence_in_one_process(room):
    from acezero_amd.session import ReconstructionSession
    seq, esd = room
    n = seq["images"].shape[0]
    opt = _opt(seq, export_point_cloud=True)
    ses = ReconstructionSession(esd, seq["images"], opt=opt, depth=seq["depth"])
    res = ses.reconstruct()
    hist = res["history"]
    rates = [h["registration_rate"] for h in hist]
    assert hist[0]["id"].startswith("iteration0_seed") and len(hist[0]["seed_rates"]) == 2
    assert 0.0 < rates[0] < 1.0 and rates[-1] >= 0.97 and hist[-1]["refit"]            # grows from a single image to the sequence
    assert all(h["iterations"] <= opt.iterations for h in hist[1:])
    ok = res["confidence"] > opt.registration_confidence
    gt = seq["poses"].cpu().numpy()
    aligned, scale = _align_similarity(res["poses"][ok].astype(np.float64), gt[ok].astype(np.float64))
    dt, ang = _pose_err(aligned, gt[ok])
    assert 0.8 < scale < 1.25                                                              # metric scale comes from the seed's depth
    assert np.median(dt) < 0.05, np.median(dt)                                             # centres within 5 cm after alignment
    rel = np.einsum("nij,njk->nik", np.linalg.inv(res["poses"][ok][:-1].astype(np.float64)), res["poses"][ok][1:].astype(np.float64))
    rel_gt = np.einsum("nij,njk->nik", np.linalg.inv(gt[ok][:-1].astype(np.float64)), gt[ok][1:].astype(np.float64))
    _, rel_ang = _pose_err(rel, rel_gt)
    assert np.median(rel_ang) < 0.5                                                        # frame-to-frame rotations
    xyz, src, sel = res["point_cloud"]
    assert len(sel) == int(ok.sum()) and xyz.shape[1] == 3 and len(xyz) == len(src) > 1000
    # the exported points lie on the room's walls (in the reconstruction's frame: map them with the same similarity)
    from acezero_amd.session import write_pose_file
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        write_pose_file(os.path.join(d, "poses_final.txt"), [f"frame_{i:04d}.png" for i in range(n)], res["poses"], res["confidence"], res["focal"])
        lines = open(os.path.join(d, "poses_final.txt")).read().splitlines()
    assert len(lines) == n and len(lines[0].split()) == 10


def test_ace_zero_script_from_image_files(tmp_path):
    """ace_zero.py's command line on files: PNG frames + 16-bit depth maps + an encoder checkpoint in, the reference's output
    files out (poses_<id>.txt / <id>.pt per round, poses_final.txt, pc_final.ply)."""
    from PIL import Image
    from acezero_amd import cli
    seq = synth.render_room_sequence(seed=7, n_frames=48, arc_deg=24.0, device="cuda")
    img = ((seq["images"][:, 0] * 0.25 + 0.4).clamp(0, 1) * 255).round().to(torch.uint8).cpu().numpy()
    dep = (seq["depth"].cpu().numpy() * 1000).round().astype(np.uint16)
    for i in range(len(img)):
        Image.fromarray(np.stack([img[i]] * 3, -1)).save(tmp_path / f"rgb_{i:04d}.png")
        Image.fromarray(np.kron(dep[i], np.ones((8, 8), np.uint16))).save(tmp_path / f"depth_{i:04d}.png")
    torch.save({k: torch.from_numpy(v) for k, v in synth.init_encoder_weights_bandpass(seed=4099).items()}, tmp_path / "encoder.pt")
    out = tmp_path / "result"
    it = "2500"
    rc = cli.ace_zero_main([str(tmp_path / "rgb_*.png"), str(out), "--depth_files", str(tmp_path / "depth_*.png"), "--encoder_path",
                            str(tmp_path / "encoder.pt"), "--use_external_focal_length", str(seq["focal"]), "--try_seeds", "1",
                            "--seed_iterations", it, "--refit_iterations", it, "--final_refit_posewait", "500", "--cooldown_iterations", "500",
                            "--iterations_max", "6", "--aug_rotation", "2", "--export_point_cloud", "True"])
    assert rc == 0
    final = open(out / "poses_final.txt").read().splitlines()
    assert len(final) == 48 and all(len(line.split()) == 10 for line in final)
    conf = np.array([float(line.split()[-1]) for line in final])
    assert (conf > 500).mean() >= 0.9
    assert (out / "poses_iteration0_seed0.txt").exists() and (out / "iteration0_seed0.pt").exists() and (out / "iteration1.pt").exists()
    head = torch.load(out / "iteration1.pt")
    assert head["fc3.weight"].dtype == torch.float16 and head["fc3.weight"].shape == (4, 512, 1, 1)               # save_model: half, Head keys
    ply = open(out / "pc_final.ply", "rb").read(200)
    assert ply.startswith(b"ply\nformat binary_little_endian")
    # export_point_cloud.py on the final network + pose file (the reference's stand-alone export)
    rc = cli.export_point_cloud_main([str(out / "pc.txt"), "--network", str(out / hist_last(out)), "--pose_file", str(out / "poses_final.txt"),
                                      "--encoder_path", str(tmp_path / "encoder.pt"), "--convention", "opencv"])
    assert rc == 0
    pts = np.loadtxt(out / "pc.txt")
    assert pts.shape[1] == 6 and len(pts) > 1000 and pts[:, 3:].min() >= 0 and pts[:, 3:].max() <= 255
    assert np.allclose(pts[:, 3], pts[:, 4]) and np.allclose(pts[:, 4], pts[:, 5])                 # grey frames -> grey points
    # the points lie on the room's walls in the reconstruction's frame: all within the room's diagonal of its centre
    assert np.percentile(np.linalg.norm(pts[:, :3] - np.median(pts[:, :3], axis=0), axis=1), 95) < 8.0


def hist_last(out):
    """The last round's head checkpoint written by ace_zero.py (iteration<k>.pt with the largest k)."""
    import re
    ks = [int(m.group(1)) for m in (re.match(r"iteration(\d+)\.pt$", f) for f in os.listdir(out)) if m]
    return f"iteration{max(ks)}.pt"


def test_train_ace_and_register_mapping_scripts_on_image_files(tmp_path):
    """train_ace.py with an ACE pose file, then register_mapping.py, both on PNG frames (the reference's two-step workflow)."""
    from PIL import Image
    from acezero_amd import cli
    seq = synth.render_room_sequence(seed=11, n_frames=40, arc_deg=20.0, device="cuda")
    img = ((seq["images"][:, 0] * 0.25 + 0.4).clamp(0, 1) * 255).round().to(torch.uint8).cpu().numpy()
    files = []
    for i in range(len(img)):
        files.append(str(tmp_path / f"rgb_{i:04d}.png"))
        Image.fromarray(np.stack([img[i]] * 3, -1)).save(files[-1])
    torch.save({k: torch.from_numpy(v) for k, v in synth.init_encoder_weights_bandpass(seed=4099).items()}, tmp_path / "encoder.pt")
    gt = seq["poses"].cpu().numpy().astype(np.float64)
    with open(tmp_path / "poses_in.txt", "w") as f:                     # every second frame is a mapping frame (confidence 2000), the rest is below the threshold
        for i in range(len(img)):
            cli.write_pose_line(f, files[i], np.linalg.inv(gt[i]), 2000 if i % 2 == 0 else 10, seq["focal"])
    out = tmp_path / "map" / "scene.pt"
    rc = cli.train_main([str(tmp_path / "rgb_*.png"), str(out), "--use_ace_pose_file", str(tmp_path / "poses_in.txt"), "--encoder_path",
                         str(tmp_path / "encoder.pt"), "--iterations", "2500", "--learning_rate_schedule", "1cyclepoly", "--learning_rate_max", "0.003",
                         "--repro_loss_type", "tanh", "--learning_rate_cooldown_iterations", "500", "--aug_rotation", "2", "--aug_scale", "1.06"])
    assert rc == 0 and out.exists()
    prelim = open(tmp_path / "map" / "poses_scene_preliminary.txt").read().splitlines()
    assert len(prelim) == 20 and prelim[0].split()[-1] == "inf"                                                   # ace_trainer.py:714
    rc = cli.register_main([str(tmp_path / "rgb_*.png"), str(out), "--encoder_path", str(tmp_path / "encoder.pt"), "--session", "query",
                            "--use_external_focal_length", str(seq["focal"]), "--hypotheses", "32", "--hypotheses_max_tries", "16"])
    assert rc == 0
    fl, poses, focals = cli.read_ace_pose_file(tmp_path / "map" / "poses_query.txt", 500)
    assert len(fl) >= 38 and np.allclose(focals, seq["focal"])
    idx = [files.index(f) for f in fl]
    dt, ang = _pose_err(poses, gt[idx])
    assert np.median(dt) < 0.01 and np.median(ang) < 0.5
    # a folder of two frame sizes (dataset.py:278-417 loads any mix): every odd frame centre-cropped to 576 columns. One context per
    # size class, random streams keyed by the position in the whole list: the uncropped frames come out exactly as above.
    mixed = tmp_path / "mixed"
    mixed.mkdir()
    w = img.shape[2]
    for i in range(len(img)):
        g = img[i] if i % 2 == 0 else img[i][:, (w - 576) // 2:(w - 576) // 2 + 576]
        Image.fromarray(np.stack([g] * 3, -1)).save(str(mixed / f"rgb_{i:04d}.png"))
    rc = cli.register_main([str(mixed / "rgb_*.png"), str(out), "--encoder_path", str(tmp_path / "encoder.pt"), "--session", "mixed",
                            "--use_external_focal_length", str(seq["focal"]), "--hypotheses", "32", "--hypotheses_max_tries", "16"])
    assert rc == 0
    rows = [l.split() for l in open(tmp_path / "map" / "poses_mixed.txt").read().splitlines()]
    ref_rows = {os.path.basename(l.split()[0]): l.split()[1:] for l in open(tmp_path / "map" / "poses_query.txt").read().splitlines()}
    assert [os.path.basename(r[0]) for r in rows] == [f"rgb_{i:04d}.png" for i in range(len(img))]
    same = 0
    for i, r in enumerate(rows):                       # (the encoder picks its tiling by chunk size: the last, shorter chunk may round differently)
        if i % 2 == 0:
            want = ref_rows[os.path.basename(r[0])]
            same += r[1:] == want
            assert np.allclose([float(x) for x in r[1:8]], [float(x) for x in want[:7]], atol=2e-3), i
    assert same >= 16
    fl, poses, _ = cli.read_ace_pose_file(tmp_path / "map" / "poses_mixed.txt", 500)
    idx = [int(os.path.basename(f)[4:8]) for f in fl]
    assert sum(1 for i in idx if i % 2) >= 17
    dt, ang = _pose_err(poses, gt[idx])
    assert np.median(dt) < 0.015 and np.median(ang) < 0.6
    # --max_estimates: the seeded subset is drawn over the whole list and written under the right names
    rc = cli.register_main([str(mixed / "rgb_*.png"), str(out), "--encoder_path", str(tmp_path / "encoder.pt"), "--session", "few",
                            "--use_external_focal_length", str(seq["focal"]), "--hypotheses", "32", "--hypotheses_max_tries", "16",
                            "--max_estimates", "10"])
    few = [l.split() for l in open(tmp_path / "map" / "poses_few.txt").read().splitlines()]
    want = np.sort(torch.randperm(len(img), generator=torch.Generator().manual_seed(1305))[:10].numpy())
    assert [os.path.basename(r[0]) for r in few] == [f"rgb_{i:04d}.png" for i in want]
    fl, poses, _ = cli.read_ace_pose_file(tmp_path / "map" / "poses_few.txt", 500)
    dt, ang = _pose_err(poses, gt[[int(os.path.basename(f)[4:8]) for f in fl]])
    assert len(fl) >= 9 and np.median(dt) < 0.015 and np.median(ang) < 0.6


def test_seed_without_usable_depth_raises_instead_of_hanging():
    """a depth map without a value in (0, 1000] never added a buffer row and the fill loop spun forever."""
    from acezero_amd.session import ReconstructionSession, check_frame_size
    seq = synth.render_room_sequence(seed=2089, n_frames=4, arc_deg=4.0, device="cuda")
    esd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights_bandpass(seed=4099).items()}
    ses = ReconstructionSession(esd, seq["images"], opt=_opt(seq), depth=torch.zeros_like(seq["depth"]))
    with pytest.raises(RuntimeError, match="depth maps have no value"):
        ses.map_seed(0, 0.3)
    with pytest.raises(RuntimeError, match="16384"):
        check_frame_size(1088, 1936)
    check_frame_size(480, 2184)


def test_reference_import_lines_train_and_register(tmp_path):
    """the reference's own import lines (`from ace_trainer import TrainerACE`, train_ace.py:20,240-241; `import
    dsacstar`, register_mapping.py:12,229) on top of the MI355X path: TrainerACE(options).train() writes the head, dsacstar.forward_rgb
    registers a frame from the scene coordinates that head predicts."""
    import dsacstar
    from ace_trainer import TrainerACE
    from PIL import Image
    from acezero_amd import cli
    from acezero_amd.network import Regressor
    seq = synth.render_room_sequence(seed=11, n_frames=24, arc_deg=12.0, device="cuda")
    img = ((seq["images"][:, 0] * 0.25 + 0.4).clamp(0, 1) * 255).round().to(torch.uint8).cpu().numpy()
    files = []
    for i in range(len(img)):
        files.append(str(tmp_path / f"rgb_{i:04d}.png"))
        Image.fromarray(np.stack([img[i]] * 3, -1)).save(files[-1])
    esd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights_bandpass(seed=4099).items()}
    torch.save(esd, tmp_path / "encoder.pt")
    gt = seq["poses"].cpu().numpy().astype(np.float64)
    with open(tmp_path / "poses_in.txt", "w") as f:
        for i in range(len(img)):
            cli.write_pose_line(f, files[i], np.linalg.inv(gt[i]), 2000, seq["focal"])
    out = tmp_path / "map" / "scene.pt"
    options = cli.train_parser().parse_args([str(tmp_path / "rgb_*.png"), str(out), "--use_ace_pose_file", str(tmp_path / "poses_in.txt"), "--encoder_path",
                                             str(tmp_path / "encoder.pt"), "--iterations", "2000", "--learning_rate_schedule", "1cyclepoly",
                                             "--learning_rate_max", "0.003", "--repro_loss_type", "tanh", "--learning_rate_cooldown_iterations", "400",
                                             "--aug_rotation", "2", "--aug_scale", "1.06"])
    trainer = TrainerACE(options)
    assert trainer.train() == 0 and out.exists()
    options.batch_size = 5000
    with pytest.raises(ValueError):
        TrainerACE(options)
    # register_mapping.py:201-242 in the reference's shape: network forward, then dsacstar.forward_rgb on a 1x3xHxW tensor
    head_sd = torch.load(out)
    net = Regressor.create_from_split_state_dict(esd, head_sd, max_frames=4, max_h=480, max_w=640)
    dsacstar.reset_call_counter(0)
    errs = []
    for i in (3, 11, 19):
        sc = net(seq["images"][i:i + 1]).float().cpu()                   # register_mapping.py:213
        out_pose = torch.zeros(4, 4)
        inliers = dsacstar.forward_rgb(sc, out_pose, 64, 10.0, seq["focal"], 320.0, 240.0, 100.0, 100.0, 8, 2089, 1000000)
        assert isinstance(inliers, int) and inliers > 1000
        errs.append(np.linalg.norm(out_pose.numpy()[:3, 3] - gt[i][:3, 3]))
    assert max(errs) < 0.02, errs

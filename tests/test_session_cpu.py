"""CPU: host logic of the in-process reconstruction session: option defaults against ace_zero.py's argparse
(tests/golden/cli_flags.json), the geometry of the on-device augmentation warp against the synthetic renderer, and that the
session refuses to run without a GPU."""
import json
import math
import os

import pytest
import torch

from acezero_amd import session, synth
from acezero_amd.encoder import output_size


def _rot_z(th):
    r = torch.eye(4)
    r[0, 0], r[0, 1], r[1, 0], r[1, 1] = math.cos(th), -math.sin(th), math.sin(th), math.cos(th)
    return r                                                         # pose_rot of dataset.py:337-343


def test_default_options_are_ace_zero_defaults(golden_dir):
    flags = json.load(open(os.path.join(golden_dir, "cli_flags.json")))
    o = vars(session.default_options())
    shared = [k for k in flags["ace_zero"] if k in o]
    assert len(shared) >= 28
    for k in shared:
        assert o[k] == flags["ace_zero"][k]["default"], k
    for k in o:                                                      # the train_ace.py flags ace_zero.py leaves at their defaults
        if k not in flags["ace_zero"] and k in flags["train_ace"]:
            assert o[k] == flags["train_ace"][k]["default"], k
    with pytest.raises(TypeError):
        session.default_options(no_such_flag=1)


@pytest.mark.parametrize("deg", [12.0, -9.0])
def test_warp_rotation_matches_camera_rotated_by_pose_rot(deg):
    """The warped frame must be what a camera with pose @ pose_rot sees: that is the pairing of image rotation and
    aug_pose_inv the training step relies on (dataset.py:324-343)."""
    th = math.radians(deg)
    seq = synth.render_room_sequence(n_frames=1, h=240, w=320, focal=262.5)
    ref = synth.render_room_sequence(h=240, w=320, focal=262.5, pose_override=(seq["poses"][0] @ _rot_z(th)).unsqueeze(0))["images"]
    for sign, lo, hi in ((1, 0.995, 1.0), (-1, -1.0, 0.7)):
        view, mask, _ = session.warp_view(seq["images"], 1.0, sign * th)
        a, b = view[0, 0][mask[0, 0]], ref[0, 0][mask[0, 0]]
        c = float(torch.corrcoef(torch.stack([a, b]))[0, 1])
        assert lo <= c <= hi, (sign, c)
    assert 0.8 < float(mask.float().mean()) < 1.0                   # corners fall outside the source frame


def test_warp_scale_matches_rendering_at_scaled_focal_and_depth_lookup():
    seq = synth.render_room_sequence(n_frames=1, h=240, w=320, focal=262.5)
    sc, th = 1.2, math.radians(11)
    hs, ws = int(240 * sc), int(320 * sc)
    ref = synth.render_room_sequence(h=hs, w=ws, focal=262.5 * hs / 240, pose_override=seq["poses"][:1])["images"]
    view, mask, _ = session.warp_view(seq["images"], sc, 0.0)
    assert view.shape == ref.shape and bool(mask.all())
    assert float(torch.corrcoef(torch.stack([view.flatten(), ref.flatten()]))[0, 1]) > 0.995
    # depth of a rotated + scaled view at its feature-map centres against the renderer's depth for that camera
    view, mask, grid = session.warp_view(seq["images"], sc, th)
    oh, ow = output_size(hs, ws)
    dv = session.view_depth(seq["depth"].unsqueeze(1), grid, oh, ow)
    gt = synth.render_room_sequence(h=hs - hs % 8, w=ws - ws % 8, focal=262.5 * hs / 240, pose_override=(seq["poses"][0] @ _rot_z(th)).unsqueeze(0))
    d_gt = gt["depth"][0]
    h2, w2 = d_gt.shape
    valid = dv[:h2, :w2] > 0
    assert float(valid.float().mean()) > 0.85
    assert float((dv[:h2, :w2] - d_gt)[valid].abs().median()) < 0.02


def test_jitter_keeps_range_and_changes_values():
    seq = synth.render_room_sequence(n_frames=1, h=64, w=96, focal=80.0)
    v0, _, _ = session.warp_view(seq["images"], 1.0, 0.0)
    v1, _, _ = session.warp_view(seq["images"], 1.0, 0.0, jitter=(1.1, 0.9))
    assert torch.allclose(v0, seq["images"], atol=1e-5)              # identity warp
    assert not torch.allclose(v0, v1) and float(v1.min()) >= (0 - 0.4) / 0.25 - 1e-5 and float(v1.max()) <= (1 - 0.4) / 0.25 + 1e-5


def test_bandpass_encoder_standin_has_reference_keys_and_zero_mean_filters():
    a, b = synth.init_encoder_weights(seed=3), synth.init_encoder_weights_bandpass(seed=3)
    assert a.keys() == b.keys() and all(a[k].shape == b[k].shape for k in a)
    for k, w in b.items():
        if k.endswith(".weight"):
            assert abs(float(w.mean(axis=(1, 2, 3)).max())) < 1e-6
        else:
            assert not w.any()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_session_needs_a_gpu():
    with pytest.raises(RuntimeError):
        session.ReconstructionSession({}, torch.zeros(1, 1, 64, 96))


def test_frame_and_depth_loaders_roundtrip(tmp_path):
    import numpy as np
    from PIL import Image
    from acezero_amd import cli
    rng = np.random.default_rng(0)
    for i in range(2):
        Image.fromarray(rng.integers(0, 256, size=(240, 320, 3), dtype=np.uint8)).save(tmp_path / f"f{i}.png")
        d = rng.integers(500, 4000, size=(30, 40)).astype(np.uint16)
        Image.fromarray(np.kron(d, np.ones((8, 8), np.uint16))).save(tmp_path / f"d{i}.png")
        if i == 1:
            last = d
    files, frames, factor = cli.load_frames(str(tmp_path / "f*.png"), image_resolution=120)
    assert [os.path.basename(f) for f in files] == ["f0.png", "f1.png"] and frames.shape == (2, 1, 120, 160) and factor == 0.5
    assert float(frames.min()) >= (0 - 0.4) / 0.25 - 1e-6 and float(frames.max()) <= (1 - 0.4) / 0.25 + 1e-6     # dataset.py:150-153
    depth = cli.load_depth_maps(str(tmp_path / "d*.png"), 2, (240, 320))
    assert depth.shape == (2, 30, 40) and np.allclose(depth[1].numpy(), last / 1000.0)                           # mm -> m at (8x+4, 8y+4)
    with pytest.raises(SystemExit):
        cli.load_depth_maps(str(tmp_path / "d0*.png"), 2, (240, 320))
    Image.fromarray(np.zeros((100, 320, 3), np.uint8)).save(tmp_path / "f2.png")
    with pytest.raises(SystemExit):
        cli.load_frames(str(tmp_path / "f*.png"), image_resolution=120)


# ------------------------------------------------------------------------------------------------------------------------
# the reconstruction loop's decisions against the reference's own ace_zero.py (tests/golden/make_ace_zero_loop_golden.py)
class _ScriptedSession(session.ReconstructionSession):
    """map / register replaced by scripted outcomes; everything else (seed trials, selection of the best seed, warm start,
    refit settings, stopping criteria, focal hand-over) is the product's reconstruct()."""

    def __init__(self, opt, rates, n=200):
        import numpy as np
        self.opt, self.n, self.H, self.W = opt, n, 480, 640
        self.depth = torch.zeros(1)
        f_ext = float(opt.use_external_focal_length)
        self.focal0 = f_ext if f_ext > 0 else -1.0          # "-1" = the heuristic, as ace_zero.py passes it on
        self.history, self.calls, self._rates, self._maps = [], [], list(rates), 0
        self._np = np
        self.rank, self.world, self.group = 0, 1, None

    def map(self, image_ids, poses_c2w, focal, *, iterations, loss_type, schedule, lr_max, refinement="none", pose_wait=0,
            refine_calibration=False, load_weights=None, with_depth=False, tag="map", data_parallel=None):
        if not with_depth:
            self._maps += 1
        self.calls.append({"cmd": "train", "id": tag, "seed": with_depth, "iterations": iterations, "loss": loss_type, "schedule": schedule,
                           "lr_max": lr_max, "pose_wait": pose_wait, "refinement": refinement, "refine_calibration": bool(refine_calibration),
                           "load_weights": None if load_weights is None else load_weights["id"], "images": len(list(image_ids))})
        return {"head": {"id": tag}, "poses_w2c": None, "focal": 500.0 + self._maps if not with_depth else focal, "iterations": iterations,
                "seconds": 0.0}

    def register(self, head_sd, focal, max_estimates=-1, tag="register", max_tries=16):
        np = self._np
        rate = self._rates.pop(0)
        conf = np.zeros(self.n, np.int32)
        conf[:round(rate * self.n)] = 1000
        self.calls.append({"cmd": "register", "network": head_sd["id"], "session": tag, "focal": focal, "max_estimates": max_estimates})
        return np.tile(np.eye(4, dtype=np.float32), (self.n, 1, 1)), conf


def _reference_decisions(calls):
    out = []
    for c in calls:
        f = c["flags"]
        if c["cmd"] == "train":
            out.append({"cmd": "train", "id": c["id"], "seed": "use_pose_seed" in f, "iterations": int(f.get("iterations", 25000)),
                        "loss": f["repro_loss_type"], "schedule": f["learning_rate_schedule"], "lr_max": float(f["learning_rate_max"]),
                        "pose_wait": int(f["pose_refinement_wait"]), "refinement": f.get("pose_refinement", "none"),
                        "refine_calibration": f.get("refine_calibration", "False") == "True",
                        "load_weights": os.path.splitext(os.path.basename(f["load_weights"]))[0] if "load_weights" in f else None})
        else:
            out.append({"cmd": "register", "network": c["network"], "session": f["session"], "focal": float(f["use_external_focal_length"]),
                        "max_estimates": int(f.get("max_estimates", -1))})
    return out


@pytest.mark.parametrize("name", ["reaches_threshold", "relative_threshold", "no_final_refine", "no_final_refit", "iterations_max",
                                  "no_warmstart", "naive_refinement_no_calibration", "slow_growth", "seed_network"])
def test_reconstruction_loop_makes_the_reference_decisions(golden_dir, name):
    g = json.load(open(os.path.join(golden_dir, "ace_zero_loop.json")))[name]
    argv = g["argv"]
    over = {argv[i].lstrip("-"): argv[i + 1] for i in range(0, len(argv), 2)}
    conv = {"final_refine": lambda v: v == "True", "final_refit": lambda v: v == "True", "warmstart": lambda v: v == "True",
            "refine_calibration": lambda v: v == "True", "iterations_max": int, "refinement": str,
            "seed_network": lambda v: {"id": "seed_network"}}       # the session takes the loaded state_dict; ace_zero.py the file
    opt = session.default_options(try_seeds=2, **{k: conv[k](v) for k, v in over.items()})
    ses = _ScriptedSession(opt, g["rates"])
    res = ses.reconstruct()
    ref = _reference_decisions(g["calls"])
    mine = [{k: v for k, v in c.items() if k != "images"} for c in ses.calls]
    assert len(mine) == len(ref), ([c.get("id") or c.get("session") for c in mine], [c.get("id") or c.get("session") for c in ref])
    for a, b in zip(mine, ref):
        assert a == b, (a, b)
    assert len(g["rates"]) - len(ses._rates) == g["registers_used"]
    # mapping rounds use the frames registered above the confidence threshold in the previous round
    trains = [c for c in ses.calls if c["cmd"] == "train" and not c["seed"]]
    first = 0 if name == "seed_network" else 2          # rates consumed before the first mapping round: none / the two seed checks
    assert [c["images"] for c in trains] == [round(r * 200) for r in g["rates"][first:first + len(trains)]]
    assert res["iterations"] == len(trains)


def test_view_mask_equals_the_zero_padded_lookup_into_a_ones_image():
    """warp_views takes the validity mask from the sampling grid (source coordinate inside (-1, W) x (-1, H)); the definition is
    dataset.py:327-328's: the mask image goes through the same warp with zero padding."""
    import numpy as np
    import torch
    rng = np.random.default_rng(3)
    img = torch.from_numpy(rng.standard_normal((3, 1, 96, 128)).astype(np.float32))
    for scale in (0.9, 1.0, 1.1):
        ang = np.radians(rng.uniform(-15, 15, size=3))
        _, mask, grid = session.warp_views(img, scale, ang)
        ref = torch.nn.functional.grid_sample(torch.ones_like(img), grid, mode="bilinear", padding_mode="zeros", align_corners=False) > 0
        assert mask.shape == ref.shape and torch.equal(mask, ref)
        assert 0.5 < float(mask.float().mean()) < 1.0

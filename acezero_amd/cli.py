"""Command-line surfaces of the hot path: same flag names and defaults as the reference's train_ace.py (:21-228)
and register_mapping.py (:47-114), driving the MI355X kernels through libacez.so.

What these entry points do NOT contain is the reference's image pipeline (dataset.py image decoding/augmentation
and the pre-trained encoder): the encoder features are taken from a file

    train_ace.py        --feature_buffer  buffer.npz     (layout of acez_train_buffer, see save_feature_buffer)
    register_mapping.py --feature_file    frames.npz     (per-frame encoder features or scene coordinates)

written by whatever fills the buffer (the reference's create_training_buffer with the 6-line hook of
INTEGRATION.md, or acezero_amd.synth for synthetic scenes). Everything after that point -- the training loop, the
schedule and early stopping, head checkpoints, pose files -- follows the reference's formats.
"""
import argparse
import logging
import math
import os
import time
from pathlib import Path

import numpy as np

from . import DEFAULT_DTYPE

_logger = logging.getLogger("acezero_amd")


def _strtobool(x):
    v = str(x).lower()
    if v in ("y", "yes", "t", "true", "on", "1"):
        return True
    if v in ("n", "no", "f", "false", "off", "0"):
        return False
    raise argparse.ArgumentTypeError(f"invalid truth value {x!r}")


# (flags, type, default, choices, help) -- train_ace.py:21-228
TRAIN_FLAGS = [
    (("--base_seed",), int, 2089, None, "seed of the derived random generators"),
    (("--pose_files",), str, None, None, "glob of per-image pose files"),
    (("--use_ace_pose_file",), Path, None, None, "ACE pose file (file qw qx qy qz tx ty tz f conf)"),
    (("--ace_pose_file_conf_threshold",), float, 1000, None, "ignore pose-file entries below this confidence"),
    (("--use_pose_seed",), float, -1, None, "map a single image with identity pose"),
    (("--depth_files",), str, None, None, "glob of depth files"),
    (("--refine_calibration",), _strtobool, False, None, "optimise the focal length during mapping"),
    (("--refine_calibration_lr",), float, 0.001, None, "learning rate of the focal-length refinement"),
    (("--use_heuristic_focal_length",), _strtobool, False, None, "use 70%% of the image diagonal as focal length"),
    (("--use_external_focal_length",), float, None, None, "externally provided focal length"),
    (("--image_resolution",), int, 480, None, "short side of the training images"),
    (("--num_data_workers",), int, 12, None, "data loader workers"),
    (("--encoder_path",), Path, "<path>", None, "pre-trained encoder weights"),
    (("--load_weights",), Path, None, None, "head weights to start from"),
    (("--num_head_blocks",), int, 1, None, "residual blocks of the head"),
    (("--use_half",), _strtobool, True, None, "16-bit matrix arithmetic (True: bf16, or fp16 with --compute_dtype fp16; False = fp32 is NOT "
                                                "implemented and is rejected, never replaced by another precision)"),
    (("--use_homogeneous",), _strtobool, True, None, "homogeneous scene-coordinate output"),
    (("--learning_rate_min",), float, 0.0005, None, ""),
    (("--learning_rate_max",), float, 0.005, None, ""),
    (("--learning_rate_schedule",), str, "circle", ["circle", "constant", "1cyclepoly"], ""),
    (("--learning_rate_warmup_iterations",), int, 1000, None, ""),
    (("--learning_rate_warmup_learning_rate",), float, 0.0005, None, ""),
    (("--learning_rate_cooldown_iterations",), int, 5000, None, ""),
    (("--learning_rate_cooldown_trigger_px_threshold",), int, 10, None, ""),
    (("--learning_rate_cooldown_trigger_percent_threshold",), float, 0.7, None, ""),
    (("--max_training_buffer_size",), int, 8000000, None, ""),
    (("--max_dataset_passes",), int, 10, None, ""),
    (("--samples_per_image",), int, 1024, None, ""),
    (("--training_buffer_cpu",), _strtobool, False, None, "accepted for compatibility; the buffer lives in HBM"),
    (("--batch_size",), int, 5120, None, ""),
    (("--iterations",), int, 25000, None, ""),
    (("--iterations_output",), int, 300, None, ""),
    (("--repro_loss_hard_clamp",), int, 1000, None, ""),
    (("--repro_loss_soft_clamp",), int, 50, None, ""),
    (("--repro_loss_soft_clamp_min",), int, 1, None, ""),
    (("--repro_loss_type",), str, "dyntanh", ["l1", "l1+sqrt", "l1+log", "tanh", "dyntanh"], ""),
    (("--repro_loss_schedule",), str, "circle", ["circle", "linear"], ""),
    (("--depth_min",), float, 0.1, None, ""),
    (("--depth_target",), float, 10, None, ""),
    (("--depth_max",), float, 1000, None, ""),
    (("--use_aug",), _strtobool, True, None, ""),
    (("--aug_rotation",), int, 15, None, ""),
    (("--aug_scale",), float, 1.5, None, ""),
    (("--render_visualization",), _strtobool, False, None, "accepted; rendering is out of scope"),
    (("--render_target_path",), Path, "renderings", None, ""),
    (("--use_existing_vis_buffer",), Path, None, None, ""),
    (("--render_flipped_portrait",), _strtobool, False, None, ""),
    (("--render_map_error_threshold",), int, 10, None, ""),
    (("--render_map_depth_filter",), int, 100, None, ""),
    (("--render_camera_z_offset",), int, 4, None, ""),
    (("--render_marker_size",), float, 0.03, None, ""),
    (("--pose_refinement",), str, "none", ["none", "naive", "mlp"], ""),
    (("--pose_refinement_weight",), float, 0.1, None, ""),
    (("--pose_refinement_wait",), int, 0, None, ""),
    (("--pose_refinement_lr",), float, 0.001, None, ""),
    (("--refinement_ortho",), str, "gram-schmidt", ["gram-schmidt", "procrustes"], ""),
]

# register_mapping.py:47-114
REGISTER_FLAGS = [
    (("--encoder_path",), Path, "<path>", None, "pre-trained encoder weights"),
    (("--session", "-sid"), None, "", None, "session name appended to the output file"),
    (("--image_resolution",), int, 480, None, ""),
    (("--num_data_workers",), int, 12, None, ""),
    (("--hypotheses", "-hyps"), int, 64, None, "RANSAC hypotheses"),
    (("--hypotheses_max_tries",), int, 1000000, None, "re-tries of an invalid minimal set"),
    (("--threshold", "-t"), float, 10, None, "inlier threshold in px"),
    (("--inlieralpha", "-ia"), float, 100, None, "soft inlier count alpha"),
    (("--maxpixelerror", "-maxerrr"), float, 100, None, "reprojection errors are clamped to this value"),
    (("--render_visualization",), _strtobool, False, None, ""),
    (("--render_target_path",), Path, "renderings", None, ""),
    (("--render_flipped_portrait",), _strtobool, False, None, ""),
    (("--render_pose_conf_threshold",), int, 5000, None, ""),
    (("--render_map_depth_filter",), int, 10, None, ""),
    (("--render_camera_z_offset",), int, 4, None, ""),
    (("--base_seed",), int, 1305, None, "torch and RANSAC seed"),
    (("--confidence_threshold",), float, 1000, None, ""),
    (("--max_estimates",), int, -1, None, "stop after this many images"),
    (("--use_external_focal_length",), float, -1, None, ""),
    (("--render_marker_size",), float, 0.03, None, ""),
]


# ace_zero.py:41-177
ACE_ZERO_FLAGS = [
    (("--depth_files",), str, None, None, "depth maps (16 bit, millimetres) for the seed images; without them the reference downloads ZoeDepth"),
    (("--iterations_max",), int, 100, None, "maximum number of mapping / relocalisation rounds"),
    (("--registration_threshold",), float, 0.99, None, "stop when this ratio of images is registered"),
    (("--relative_registration_threshold",), float, 0.01, None, "stop when fewer new images than this were registered"),
    (("--final_refine",), _strtobool, True, None, "one more mapping round after the stopping criteria are met"),
    (("--final_refit",), _strtobool, True, None, "refit a fresh network in the last round"),
    (("--final_refit_posewait",), int, 5000, None, ""),
    (("--refit_iterations",), int, 25000, None, ""),
    (("--registration_confidence",), int, 500, None, "inlier count above which an image counts as registered"),
    (("--try_seeds",), int, 5, None, "number of seed images to try"),
    (("--seed_parallel_workers",), int, 3, None, "accepted; seeds run back to back in one process"),
    (("--seed_iterations",), int, 10000, None, ""),
    (("--seed_network",), Path, None, None, "pre-trained head to start from"),
    (("--warmstart",), _strtobool, True, None, ""),
    (("--export_point_cloud",), _strtobool, False, None, ""),
    (("--dense_point_cloud",), _strtobool, False, None, ""),
    (("--refinement",), str, "mlp", ["mlp", "none", "naive"], ""),
    (("--refinement_ortho",), str, "gram-schmidt", ["gram-schmidt", "procrustes"], ""),
    (("--pose_refinement_wait",), int, 0, None, ""),
    (("--pose_refinement_lr",), float, 0.001, None, ""),
    (("--refine_calibration",), _strtobool, True, None, ""),
    (("--use_external_focal_length",), float, -1, None, "-1: 70%% of the image diagonal"),
    (("--learning_rate_schedule",), str, "1cyclepoly", ["circle", "constant", "1cyclepoly"], ""),
    (("--learning_rate_max",), float, 0.003, None, ""),
    (("--cooldown_iterations",), int, 5000, None, ""),
    (("--cooldown_threshold",), float, 0.7, None, ""),
    (("--image_resolution",), int, 480, None, ""),
    (("--num_head_blocks",), int, 1, None, ""),
    (("--max_dataset_passes",), int, 10, None, ""),
    (("--repro_loss_type",), str, "tanh", ["l1", "l1+sqrt", "l1+log", "tanh", "dyntanh"], ""),
    (("--repro_loss_hard_clamp",), int, 1000, None, ""),
    (("--repro_loss_soft_clamp",), int, 50, None, ""),
    (("--aug_rotation",), int, 15, None, ""),
    (("--num_data_workers",), int, 12, None, "accepted; frames are decoded once by the main process"),
    (("--training_buffer_cpu",), _strtobool, False, None, "accepted; the buffer lives in HBM"),
    (("--ransac_iterations",), int, 32, None, ""),
    (("--ransac_threshold",), float, 10, None, ""),
    (("--render_visualization",), _strtobool, False, None, "accepted; rendering is out of scope"),
    (("--render_flipped_portrait",), _strtobool, False, None, ""),
    (("--render_marker_size",), float, 0.03, None, ""),
    (("--iterations_output",), int, 500, None, ""),
    (("--random_seed",), int, 1305, None, ""),
]


def _add(parser, table):
    for flags, typ, default, choices, hlp in table:
        kw = {"default": default, "help": hlp}
        if typ is not None:
            kw["type"] = typ
        if choices is not None:
            kw["choices"] = choices
        parser.add_argument(*flags, **kw)


def _add_dtype(p):
    p.add_argument("--compute_dtype", default=None, choices=["bf16", "fp16"],
                   help="[additive] 16-bit operand format of encoder and head (fp32 accumulation in both): fp16 is the reference's autocast "
                        "arithmetic (ace_trainer.py:366-367,517-518, register_mapping.py:209-210), bf16 what BASELINE.json's north_star "
                        "names; default: $ACEZ_DTYPE, else " + DEFAULT_DTYPE)


def train_parser():
    p = argparse.ArgumentParser(description="Fast training of a scene coordinate regression network (MI355X head trainer).",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("rgb_files", type=str, help="glob of the RGB files (recorded in the outputs; pixels are not read here)")
    p.add_argument("output_map_file", type=Path, help="target file for the trained head")
    _add(p, TRAIN_FLAGS)
    p.add_argument("--feature_buffer", type=Path, default=None, help="[additive] .npz training buffer (acez_train_buffer layout)")
    _add_dtype(p)
    p.add_argument("--num_gpus", type=int, default=1, help="[additive] informational; multi-GPU reconstructions are launched as "
                   "`torchrun --nproc-per-node G ace_zero.py ...` (the mapping rounds inside are data parallel)")
    return p


def register_parser():
    p = argparse.ArgumentParser(description="Estimate camera poses for a set of images (MI355X DSAC*).",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("rgb_files", type=str, help="glob of the RGB files")
    p.add_argument("network", type=Path, help="head weights of the scene")
    _add(p, REGISTER_FLAGS)
    p.add_argument("--feature_file", type=Path, default=None, help="[additive] .npz with per-frame encoder features or scene coordinates")
    _add_dtype(p)
    return p


def ace_zero_parser():
    p = argparse.ArgumentParser(description="Run ACE0 for a scene in ONE process on one MI355X (acezero_amd.session).",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("rgb_files", type=str, help="glob of the RGB files, e.g. 'datasets/scene/*.jpg'")
    p.add_argument("results_folder", type=Path, help="output folder")
    _add(p, ACE_ZERO_FLAGS)
    p.add_argument("--encoder_path", type=Path, default=Path(__file__).resolve().parent.parent / "ace_encoder_pretrained.pt",
                   help="[additive] pre-trained encoder weights (train_ace.py / register_mapping.py take the same flag)")
    _add_dtype(p)
    return p


def export_point_cloud_parser():
    """export_point_cloud.py:26-58."""
    p = argparse.ArgumentParser(description="Extract point cloud from network or visualization buffer file; .txt and .ply are supported.",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("output_file", type=Path)
    p.add_argument("--network", type=Path, help="network to extract point cloud from.")
    p.add_argument("--pose_file", type=Path, help="pose file of images that trained the network")
    p.add_argument("--visualization_buffer", type=Path, help="vis buffer file that contains a pre-calculated point cloud.")
    p.add_argument("--encoder_path", type=Path, default="<path>", help="file containing pre-trained encoder weights")
    p.add_argument("--image_resolution", type=int, default=480, help="base image resolution")
    p.add_argument("--confidence_threshold", type=int, default=500)
    p.add_argument("--convention", type=str, default="opengl", choices=["opengl", "opencv"], help="coordinate convention of the point cloud")
    p.add_argument("--dense_point_cloud", type=_strtobool, default=False, help="do not filter points based on reprojection error")
    _add_dtype(p)
    return p


def _default_encoder_path(path):
    return Path(__file__).resolve().parent.parent / "ace_encoder_pretrained.pt" if str(path) == "<path>" else Path(path)


# ------------------------------------------------------------------------------------------------------- file formats
def write_pose_line(f, rgb_file, pose_w2c, confidence, focal_length):
    """dataset_io.py:159-186: `file qw qx qy qz tx ty tz focal confidence`, world->cam."""
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(np.asarray(pose_w2c, np.float64)[:3, :3]).as_quat()
    t = np.asarray(pose_w2c)[:3, 3]
    f.write(f"{rgb_file} {q[3]} {q[0]} {q[1]} {q[2]} {t[0]} {t[1]} {t[2]} {focal_length} {confidence}\n")


def save_feature_buffer(path, prob, image_files=None, with_depth_targets=False):
    """Write a training buffer (dict in the layout of acezero_amd.synth.make_training_problem) as .npz."""
    n_img = prob["image_pose_inv"].shape[0]
    files = image_files if image_files is not None else [f"frame_{i:06d}.png" for i in range(n_img)]
    np.savez(path, features=prob["features"].astype(np.float16), target_px=prob["target_px"], view_idx=prob["view_idx"],
             view_aug_inv=prob["view_aug_inv"], view_K=prob["view_K"], view_Kinv=prob["view_Kinv"], view_image=prob["view_image"],
             image_pose_inv=prob["image_pose_inv"], mean=prob["mean"], focal=np.float32(prob["focal"]), image_files=np.array(files),
             **({"target_crds": prob["target_crds"]} if with_depth_targets else {}))


# ------------------------------------------------------------------------------------------------------------ train
def train_main(argv=None):
    opt = train_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    return train_with_options(opt)


def train_with_options(opt):
    """The body of train_ace.py after argument parsing: what the reference runs as TrainerACE(options).train() (train_ace.py:240-241;
    the top-level ace_trainer.TrainerACE shim calls this)."""
    import torch
    from .head import HeadTrainer
    if opt.batch_size % 512 != 0:
        raise SystemExit("batch_size must be a multiple of 512 (train_ace.py:138)")
    if not opt.use_half:
        raise SystemExit("--use_half False (fp32 head arithmetic, ace_trainer.py:330) is not implemented on this path; it is refused rather "
                         "than silently run in 16 bits. Use --use_half True [--compute_dtype fp16 for the reference's autocast precision].")
    dtype = getattr(opt, "compute_dtype", None)
    if opt.feature_buffer is None:
        return _train_from_images(opt)   # (the session's encoder and head take opt.compute_dtype: _session_options)
    buf = np.load(opt.feature_buffer, allow_pickle=False)
    n = min(int(buf["features"].shape[0]), opt.max_training_buffer_size)
    focal = float(opt.use_external_focal_length) if opt.use_external_focal_length is not None else float(buf["focal"])
    tr = HeadTrainer(buf["mean"], num_head_blocks=opt.num_head_blocks, use_homogeneous=opt.use_homogeneous, max_batch=opt.batch_size,
                     loss_type=opt.repro_loss_type, soft_clamp=opt.repro_loss_soft_clamp, soft_clamp_min=opt.repro_loss_soft_clamp_min,
                     circle_schedule=opt.repro_loss_schedule == "circle", hard_clamp=opt.repro_loss_hard_clamp, depth_min=opt.depth_min,
                     depth_max=opt.depth_max, depth_target=opt.depth_target,
                     inlier_px_threshold=opt.learning_rate_cooldown_trigger_px_threshold, schedule=opt.learning_rate_schedule,
                     iterations=opt.iterations, lr_min=opt.learning_rate_min, lr_max=opt.learning_rate_max,
                     warmup_iterations=opt.learning_rate_warmup_iterations, warmup_lr=opt.learning_rate_warmup_learning_rate,
                     cooldown_iterations=opt.learning_rate_cooldown_iterations,
                     cooldown_trigger_percent=opt.learning_rate_cooldown_trigger_percent_threshold,
                     refine_calibration=opt.refine_calibration, focal_init=focal, calib_lr=opt.refine_calibration_lr,
                     pose_refinement=opt.pose_refinement, pose_refinement_wait=opt.pose_refinement_wait,
                     pose_refinement_lr=opt.pose_refinement_lr, pose_refinement_weight=opt.pose_refinement_weight,
                     refinement_ortho=opt.refinement_ortho,
                     pose_seed=opt.base_seed + 511, initial_poses=buf["image_pose_inv"][:, :3] if opt.pose_refinement == "naive" else None,
                     dtype=dtype)
    if opt.load_weights is not None:
        tr.load_state_dict(torch.load(opt.load_weights, map_location="cpu"))
        _logger.info(f"Loaded weights from: {opt.load_weights}")
    else:
        g = torch.Generator().manual_seed(opt.base_seed + 1023)     # ace_trainer.py:66-69 network-initialisation generator
        bound = 1.0 / math.sqrt(512.0)
        tr.load_flat((torch.rand(tr.n_params, generator=g) * 2 - 1) * bound)
    tr.set_buffer(buf["features"][:n].astype(np.float32), buf["target_px"][:n], buf["view_idx"][:n], buf["view_aug_inv"], buf["view_K"],
                  buf["view_Kinv"], buf["view_image"], buf["image_pose_inv"],
                  target_crds=buf["target_crds"][:n] if "target_crds" in buf.files else None)   # present = use_depth (ace_trainer.py:86-92)
    _logger.info(f"Training buffer: {n} patches, {buf['image_pose_inv'].shape[0]} images.")

    log_path = opt.output_map_file.with_suffix(".txt")
    start = time.time()
    epoch, launched, done = 0, 0, False
    orig_poses = np.linalg.inv(buf["image_pose_inv"].astype(np.float64))[:, :3, 3]
    with open(log_path, "w", 1) as log:
        from .head import epoch_batches
        pairs = epoch_batches(n, opt.batch_size, opt.base_seed + 8191, tr.device)   # ace_trainer.py:79-80,466 (drawn on the device)
        while not done:
            for _ in range(n // opt.batch_size):
                tr.step(*next(pairs))                                # (rows, next rows): the next batch is gathered ahead
                launched += 1
                if launched % opt.iterations_output == 0 or launched % 64 == 0:
                    st = tr.state()                                  # the only host synchronisation of the loop
                    if st["nan"]:
                        raise SystemExit("Aborting because of NaN loss")          # ace_trainer.py:615-617
                    if launched % opt.iterations_output == 0:
                        it = st["iteration"] - 1
                        elapsed = time.time() - start
                        _logger.info(f"Iteration: {it:6d}|{st['max_iterations']:6d} / Epoch {epoch:03d}, Loss: {st['loss']:.1f}, "
                                     f"Batch inliers ({opt.learning_rate_cooldown_trigger_px_threshold}px): "
                                     f"{st['batch_inliers'] * 100:.1f}%, Time: {elapsed:.0f}s")
                        cur = np.linalg.inv(np.concatenate([tr.current_poses(), np.tile([[[0, 0, 0, 1.0]]], (len(orig_poses), 1, 1))], 1))[:, :3, 3]
                        d = np.linalg.norm(cur - orig_poses, axis=1)
                        row = f"{it} {elapsed} {st['loss']} {st['batch_inliers']} {d.mean()} {d.min()} {d.max()}"
                        if opt.refine_calibration:
                            row += f" {st['focal_scale'] * focal}"
                        log.write(row + "\n")
                    if st["iteration"] >= st["max_iterations"]:
                        done = True
                        break
            epoch += 1
    st = tr.state()
    elapsed = time.time() - start
    _logger.info(f"Done without errors. Training time: {elapsed:.1f}s, {st['iteration']} iterations, "
                 f"{st['iteration'] * opt.batch_size / max(elapsed, 1e-9):.0f} patches/s.")
    # save_model (ace_trainer.py:681-694): half-precision head state_dict
    opt.output_map_file.parent.mkdir(parents=True, exist_ok=True)
    torch.save({k: v.detach().cpu().half() for k, v in tr.state_dict().items()}, opt.output_map_file)
    # save_poses (ace_trainer.py:696-728)
    pose_file = opt.output_map_file.parent / f"poses_{opt.output_map_file.stem}_preliminary.txt"
    files = [str(x) for x in buf["image_files"]] if "image_files" in buf.files else [f"{i}" for i in range(len(orig_poses))]
    f_out = st["focal_scale"] * focal if opt.refine_calibration else focal
    with open(pose_file, "w") as f:
        for i, p34 in enumerate(tr.current_poses()):
            write_pose_line(f, files[i], p34, float("inf"), f_out)
    _logger.info(f"Saved trained head weights to: {opt.output_map_file}; refined poses to: {pose_file}")
    return 0


def read_ace_pose_file(path, confidence_threshold):
    """dataset_io.load_dataset_ace (:96-156): (files, cam->world 4x4 float64 [k,4,4], focal lengths) of the entries whose confidence
    is not below the threshold."""
    from scipy.spatial.transform import Rotation
    files, poses, focals = [], [], []
    for line in open(path).read().splitlines():
        tok = line.split()
        assert len(tok) == 10, f"Expected 10 tokens per line in pose file, got {len(tok)}"
        if float(tok[-1]) < confidence_threshold:
            continue
        q = [float(t) for t in tok[1:5]]
        T = np.eye(4)
        T[:3, :3] = Rotation.from_quat(q[1:] + [q[0]]).as_matrix()
        T[:3, 3] = [float(t) for t in tok[5:8]]
        files.append(tok[0]); poses.append(np.linalg.inv(T)); focals.append(float(tok[-2]))
    return files, np.stack(poses) if poses else np.zeros((0, 4, 4)), focals


def _session_options(opt, **extra):
    from .session import default_options
    known = vars(default_options())
    over = {k: v for k, v in vars(opt).items() if k in known and v is not None}
    over.update(extra)
    return default_options(**over)


def _train_from_images(opt):
    """train_ace.py on image files: poses from --use_ace_pose_file / --pose_files / --use_pose_seed, one mapping run of the session."""
    import glob
    import torch
    from .session import ReconstructionSession
    if opt.use_ace_pose_file is not None:
        files, poses, focals = read_ace_pose_file(opt.use_ace_pose_file, opt.ace_pose_file_conf_threshold)
        files, frames, fscale = load_frames(None, opt.image_resolution, files=files)
    else:
        files, frames, fscale = load_frames(opt.rgb_files, opt.image_resolution)
        focals = []
        poses = np.stack([np.loadtxt(f) for f in sorted(glob.glob(opt.pose_files))]) if opt.pose_files is not None else None   # dataset_io.load_pose
    depth = load_depth_maps(opt.depth_files, len(files), frames.shape[2:]) if opt.depth_files is not None else None
    ids = list(range(len(files)))
    if opt.use_pose_seed >= 0:                                           # dataset.py:110-124
        ids, poses = [int(opt.use_pose_seed * len(files))], np.eye(4)[None]
        if depth is None:
            raise SystemExit("--use_pose_seed needs --depth_files here (the reference's ZoeDepth fallback is a network download)")
    elif poses is None or len(poses) != len(files):
        raise SystemExit("need one pose per image: --use_ace_pose_file, --pose_files or --use_pose_seed")
    H, W = frames.shape[2:]
    if opt.use_external_focal_length is not None:
        focal = opt.use_external_focal_length * fscale
    elif opt.use_heuristic_focal_length or not focals:
        focal = math.sqrt(W ** 2 + H ** 2) * 0.7
    else:
        assert np.allclose(focals, focals[0]), "a single focal length is supported"
        focal = focals[0] * fscale
    so = _session_options(opt, cooldown_iterations=opt.learning_rate_cooldown_iterations, use_external_focal_length=focal,
                          cooldown_threshold=opt.learning_rate_cooldown_trigger_percent_threshold)
    ses = ReconstructionSession(torch.load(_default_encoder_path(opt.encoder_path), map_location="cpu"), frames, opt=so, depth=depth)
    m = ses.map(ids, torch.from_numpy(np.asarray(poses, np.float64)), focal, iterations=opt.iterations, loss_type=opt.repro_loss_type,
                schedule=opt.learning_rate_schedule, lr_max=opt.learning_rate_max, refinement=opt.pose_refinement,
                pose_wait=opt.pose_refinement_wait, refine_calibration=opt.refine_calibration,
                load_weights=torch.load(opt.load_weights, map_location="cpu") if opt.load_weights is not None else None,
                with_depth=opt.use_pose_seed >= 0 or opt.depth_files is not None, tag=opt.output_map_file.stem)
    opt.output_map_file.parent.mkdir(parents=True, exist_ok=True)
    torch.save(m["head"], opt.output_map_file)                           # save_model (ace_trainer.py:681-694)
    pose_file = opt.output_map_file.parent / f"poses_{opt.output_map_file.stem}_preliminary.txt"
    with open(pose_file, "w") as f:                                      # save_poses (ace_trainer.py:696-728)
        for j, i in enumerate(ids):
            write_pose_line(f, files[i], np.vstack([m["poses_w2c"][j], [0, 0, 0, 1.0]]), float("inf"), m["focal"] / fscale)
    _logger.info(f"Done without errors. {m['iterations']} iterations in {m['seconds']:.1f}s ({m['patches_per_s']:.0f} patches/s). "
                 f"Saved trained head weights to: {opt.output_map_file}; refined poses to: {pose_file}")
    return 0


def frame_size_classes(rgb_glob):
    """The sorted file list grouped by image size (header reads only): {(W, H): [positions in the sorted list]}. One size means
    one resize factor and one resized shape for load_frames. The reference's batch-size-1 loaders take any mix (dataset.py:278-417)."""
    import glob
    from PIL import Image
    files = sorted(glob.glob(rgb_glob))
    if not files:
        raise SystemExit(f"no files match {rgb_glob!r}")
    classes = {}
    for i, f in enumerate(files):
        with Image.open(f) as im:
            classes.setdefault(im.size, []).append(i)
    return files, classes


def _register_mixed_sizes(opt, files, classes):
    """register_mapping.py on a folder whose frames have several sizes: registration is independent per frame, so every size class
    gets its own encoder / RANSAC context (one ReconstructionSession each); --max_estimates draws its seeded subset over the whole
    list, the random streams are keyed by the position in the whole list, the pose file keeps the list's order."""
    import torch
    from .session import ReconstructionSession
    n = len(files)
    if opt.max_estimates <= 0 or opt.max_estimates >= n:
        chosen = np.arange(n)
    else:
        chosen = np.sort(torch.randperm(n, generator=torch.Generator().manual_seed(int(opt.base_seed)))[:opt.max_estimates].numpy())
    keep = set(int(i) for i in chosen)
    enc_sd = torch.load(_default_encoder_path(opt.encoder_path), map_location="cpu")
    head_sd = torch.load(opt.network, map_location="cpu")
    rows = {}
    for (w, h), pos in sorted(classes.items()):
        pos = [i for i in pos if i in keep]
        if not pos:
            continue
        sub_files, frames, fscale = load_frames(None, opt.image_resolution, files=[files[i] for i in pos])
        so = _session_options(opt, use_external_focal_length=opt.use_external_focal_length * fscale if opt.use_external_focal_length > 0 else -1.0,
                              ransac_iterations=opt.hypotheses, ransac_threshold=opt.threshold, register_seed=opt.base_seed, use_aug=False,
                              registration_confidence=opt.confidence_threshold)
        ses = ReconstructionSession(enc_sd, frames, opt=so)
        poses, inl = ses.register(head_sd, ses.focal0, max_tries=opt.hypotheses_max_tries, rng_ids=pos, tag=f"register {w}x{h}")
        for k, i in enumerate(pos):
            rows[i] = (poses[k], int(inl[k]), ses.focal0 / fscale)
        del ses
    out = Path(opt.network).parent / f"poses_{opt.session}.txt"
    with open(out, "w") as f:
        for i in sorted(rows):
            p, c, focal = rows[i]
            write_pose_line(f, files[i], np.linalg.inv(np.asarray(p, np.float64)), c, float(focal))
    _logger.info(f"Registered {len(rows)} images of {len(classes)} sizes -> {out}")
    return 0


def _register_from_images(opt):
    """register_mapping.py on image files: encoder -> head -> RANSAC for every frame (register_mapping.py:201-276)."""
    import torch
    from .session import ReconstructionSession, write_pose_file
    all_files, classes = frame_size_classes(opt.rgb_files)
    if len(classes) > 1:
        return _register_mixed_sizes(opt, all_files, classes)
    files, frames, fscale = load_frames(opt.rgb_files, opt.image_resolution)
    so = _session_options(opt, use_external_focal_length=opt.use_external_focal_length * fscale if opt.use_external_focal_length > 0 else -1.0,
                          ransac_iterations=opt.hypotheses, ransac_threshold=opt.threshold, register_seed=opt.base_seed, use_aug=False,
                          registration_confidence=opt.confidence_threshold)
    ses = ReconstructionSession(torch.load(_default_encoder_path(opt.encoder_path), map_location="cpu"), frames, opt=so)
    poses, inl = ses.register(torch.load(opt.network, map_location="cpu"), ses.focal0, max_estimates=opt.max_estimates, max_tries=opt.hypotheses_max_tries)
    out = Path(opt.network).parent / f"poses_{opt.session}.txt"
    write_pose_file(out, [files[i] for i in ses.registered_ids], poses, inl, ses.focal0 / fscale)
    _logger.info(f"Registered {len(poses)} images -> {out}")
    return 0


# --------------------------------------------------------------------------------------------------------- register
def register_main(argv=None):
    import torch
    from . import dsacstar
    from .head import HeadTrainer
    opt = register_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    if opt.feature_file is None:
        return _register_from_images(opt)
    torch.manual_seed(opt.base_seed)
    data = np.load(opt.feature_file, allow_pickle=False)
    files = [str(x) for x in data["image_files"]]
    # --max_estimates: a seeded random subset, as the reference's DataLoader(shuffle=True) under torch.manual_seed(base_seed) gives
    # (register_mapping.py:122-147,256), in file order -- not the first n frames
    if opt.max_estimates <= 0 or opt.max_estimates >= len(files):
        ids = np.arange(len(files))
    else:
        ids = np.sort(torch.randperm(len(files), generator=torch.Generator().manual_seed(int(opt.base_seed)))[:opt.max_estimates].numpy())
    n = len(ids)
    t0 = time.time()
    if "scene_coordinates" in data.files:
        sc = torch.from_numpy(data["scene_coordinates"][ids].astype(np.float32)).cuda()
    else:
        sd = torch.load(opt.network, map_location="cpu")
        nb = sum(1 for k in sd if k.endswith("c0.weight"))
        head = HeadTrainer(sd["mean"].float().view(3), num_head_blocks=nb, use_homogeneous=sd["fc3.weight"].shape[0] == 4, max_batch=8192,
                           iterations=1, inference_only=True, dtype=opt.compute_dtype)
        head.load_state_dict(sd)                                    # fp16 checkpoint -> fp32 masters -> 16-bit compute copies
        h, w = int(data["h"]), int(data["w"])
        feats = torch.from_numpy(data["features"][ids].astype(np.float32)).cuda().reshape(-1, 512)
        sc = head.get_scene_coordinates(feats).reshape(n, h, w, 3).permute(0, 3, 1, 2).contiguous()
    f_ext = opt.use_external_focal_length
    focal = np.broadcast_to(np.asarray(data["focal"], np.float32), (len(files),)) if f_ext < 0 else np.full(len(files), f_ext, np.float32)
    ppx = np.broadcast_to(np.asarray(data["ppx"], np.float32), (len(files),))
    ppy = np.broadcast_to(np.asarray(data["ppy"], np.float32), (len(files),))
    prm = dict(hyps=opt.hypotheses, thr=opt.threshold, alpha=opt.inlieralpha, max_reproj=opt.maxpixelerror, sub=8, max_tries=opt.hypotheses_max_tries)
    poses, inl, _ = dsacstar.register_batch(sc, [(focal[i], ppx[i], ppy[i]) for i in ids], prm, opt.base_seed, [int(i) for i in ids],
                                            want_masks=False)
    poses, inl = poses.cpu().numpy(), inl.cpu().numpy()
    out_dir = Path(opt.network).parent
    pose_log_file = out_dir / f"poses_{opt.session}.txt"
    with open(pose_log_file, "w") as f:
        for k, i in enumerate(ids):
            _logger.info(f"Frame: {files[i]}, Confidence: {int(inl[k])}")
            write_pose_line(f, files[i], np.linalg.inv(poses[k].astype(np.float64)), int(inl[k]), float(focal[i]))   # :261-276
    dt = time.time() - t0
    _logger.info(f"Registered {n} images in {dt:.2f}s ({n / max(dt, 1e-9):.0f} images/s) -> {pose_log_file}")
    return 0


# --------------------------------------------------------------------------------------------------------- ace_zero
def load_frames(rgb_glob, image_resolution=480, files=None, return_rgb=False):
    """Minimal stand-in for CamLocDataset's image path without augmentation (dataset.py:189-195,227-237,146-160): decode,
    resize so that the short side is `image_resolution` (PIL bilinear, as torchvision does for PIL images), grey, normalise.
    All frames must have one size (the session batches them). Returns (files, float32 [n,1,H,W], resize factor): focal lengths
    on the command line and in pose files refer to the ORIGINAL image size and are multiplied by that factor (dataset.py:289-290)."""
    import glob
    import torch
    from PIL import Image
    if files is None:
        files = sorted(glob.glob(rgb_glob))                             # dataset_io.get_files_from_glob sorts
    if not files:
        raise SystemExit(f"no files match {rgb_glob!r}")
    frames, size, factor, rgbs = [], None, 1.0, []
    for f in files:
        im = Image.open(f).convert("RGB")
        w, h = im.size
        sc = image_resolution / min(w, h)
        nw, nh = (image_resolution, int(h * sc)) if w <= h else (int(w * sc), image_resolution)
        small = im.resize((nw, nh), Image.BILINEAR)
        g = np.asarray(small.convert("L"), np.float32) / 255.0
        if return_rgb:
            rgbs.append(np.asarray(small, np.uint8))
        if size is None:
            size = g.shape
            from .session import check_frame_size
            try:
                check_frame_size(*size)                                  # before the other frames are decoded
            except RuntimeError as e:
                raise SystemExit(str(e))
        elif g.shape != size:
            raise SystemExit(f"{f}: resized frame is {g.shape}, the first one {size}: the in-process mapping session batches frames of ONE "
                             "size (register_mapping.py handles folders of mixed sizes, one context per size class; the reference's "
                             "batch-size-1 loaders accept them everywhere): crop or pad the images to a common aspect ratio for mapping")
        frames.append((g - 0.4) / 0.25)
        factor = sc
    if return_rgb:
        return files, torch.from_numpy(np.stack(frames)[:, None]), factor, np.stack(rgbs)
    return files, torch.from_numpy(np.stack(frames)[:, None]), factor


def load_depth_maps(depth_glob, n, frame_hw):
    """--depth_files (dataset.py:299-304,333,359): 16-bit millimetres -> metres, nearest resize to the frame, value at the
    feature-map pixel centres (offset 4, stride 8). Returns float32 [n, ceil(H/8), ceil(W/8)]."""
    import glob
    import torch
    from PIL import Image
    files = sorted(glob.glob(depth_glob))
    if len(files) != n:
        raise SystemExit(f"{len(files)} depth files for {n} images")
    H, W = frame_hw
    out = np.zeros((n, (H + 7) // 8, (W + 7) // 8), np.float32)
    for i, f in enumerate(files):
        d = np.asarray(Image.open(f).resize((W, H), Image.NEAREST), np.float32) / 1000.0
        sub = d[4::8, 4::8]
        out[i, :sub.shape[0], :sub.shape[1]] = sub
    return torch.from_numpy(out)


def ace_zero_main(argv=None):
    import torch
    from .session import ReconstructionSession, default_options, write_pose_file
    opt = ace_zero_parser().parse_args(argv)
    # `torchrun --nproc-per-node G ace_zero.py ...`: one process per GPU (RCCL). Frames, buffer and registration are sharded inside
    # the session; rank 0 writes the files. Without torchrun this is the single-GPU run.
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if not dist.is_initialized():
            dist.init_process_group(os.environ.get("ACEZ_DIST_BACKEND", "nccl"))
        if opt.export_point_cloud:
            raise SystemExit("--export_point_cloud True runs on one GPU: export from the written pose file with export_point_cloud.py")
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING)
    opt.results_folder.mkdir(parents=True, exist_ok=True)
    files, frames, fscale, rgb = load_frames(opt.rgb_files, opt.image_resolution, return_rgb=True)
    depth = load_depth_maps(opt.depth_files, len(files), frames.shape[2:]) if opt.depth_files is not None else None
    if depth is None and opt.seed_network is None:
        raise SystemExit("ace_zero.py (MI355X): seeds need --depth_files (or --seed_network); the reference's ZoeDepth fallback is a "
                         "network download and not part of this package")
    known = vars(default_options())
    over = {k: v for k, v in vars(opt).items() if k in known and k != "seed_network"}
    if opt.use_external_focal_length > 0:
        over["use_external_focal_length"] = opt.use_external_focal_length * fscale
    if opt.seed_network is not None:
        over["seed_network"] = torch.load(opt.seed_network, map_location="cpu")
    ses = ReconstructionSession(torch.load(_default_encoder_path(opt.encoder_path), map_location="cpu"), frames, opt=default_options(**over), depth=depth)
    res = ses.reconstruct()
    if rank != 0:                                                       # every rank holds the same result; rank 0 writes it
        import torch.distributed as dist
        dist.barrier()
        return 0
    for h in res["history"]:                                            # the files ace_zero.py leaves behind (SURVEY 8b "process/file contract")
        write_pose_file(opt.results_folder / f"poses_{h['id']}.txt", files, h["poses"], h["confidence"], h["focal"] / fscale)
        torch.save(h["head"], opt.results_folder / f"{h['id']}.pt")
        _logger.info(f"{h['id']}: registered {h['registration_rate'] * 100:.1f}% of the images")
    write_pose_file(opt.results_folder / "poses_final.txt", files, res["poses"], res["confidence"], res["focal"] / fscale)
    if opt.export_point_cloud:
        from .pointcloud import write_point_cloud
        xyz, src, sel = res["point_cloud"]
        f, p = np.divmod(src.astype(np.int64), ses.hw)
        write_point_cloud(opt.results_folder / "pc_final.ply", xyz, source_colours(rgb, sel[f], p, ses.ow))
    rates = [float((res["confidence"] > t).mean()) for t in (500, 1000, 2000, 4000)]
    _logger.info(f"Reconstructed in {res['seconds'] / 60:.1f} minutes, {res['iterations']} iterations; "
                 "registration rate @500/@1000/@2000/@4000: " + " ".join(f"{r * 100:.1f}%" for r in rates))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    return 0


# ------------------------------------------------------------------------------------------------ export_point_cloud
def source_colours(rgb_nhw3, frame_of_point, pixel_of_point, map_w):
    """Colour of every kept map pixel: the image value at its centre (nearest-neighbour sub-sampling with offset 4, stride 8,
    ace_vis_util.py:566-570), as 0..255 floats."""
    y, x = np.divmod(pixel_of_point.astype(np.int64), map_w)
    yy = np.minimum(y * 8 + 4, rgb_nhw3.shape[1] - 1)
    xx = np.minimum(x * 8 + 4, rgb_nhw3.shape[2] - 1)
    return rgb_nhw3[frame_of_point, yy, xx].astype(np.float64)


def export_point_cloud_main(argv=None):
    """export_point_cloud.py: from a visualisation buffer (host only) or from network + pose file (encoder -> head -> filter on
    the device)."""
    import pickle
    import torch
    from .pointcloud import write_point_cloud
    parser = export_point_cloud_parser()
    opt = parser.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    if opt.visualization_buffer is None and (opt.network is None or opt.pose_file is None):
        parser.error("You must provide either a visualization buffer or network and pose file.")
    if opt.dense_point_cloud and opt.visualization_buffer is not None:
        parser.error("A dense cloud cannot be extracted from a visualization buffer. Please provide network and pose file.")
    if opt.visualization_buffer is not None:
        with open(opt.visualization_buffer, "rb") as f:
            state = pickle.load(f)
        xyz, clr = np.asarray(state["map_xyz"]).copy(), np.asarray(state["map_clr"])
        if opt.convention == "opencv":                                   # the buffer holds OpenGL coordinates (:104-107)
            xyz[:, 1], xyz[:, 2] = -xyz[:, 1], -xyz[:, 2]
    else:
        from .session import ReconstructionSession, default_options
        files, c2w, focals = read_ace_pose_file(opt.pose_file, opt.confidence_threshold)
        if not files:
            raise SystemExit("no pose above the confidence threshold")
        assert np.allclose(focals, focals[0]), "a single focal length is supported"
        files, frames, fscale, rgb = load_frames(None, opt.image_resolution, files=files, return_rgb=True)
        so = default_options(use_external_focal_length=focals[0] * fscale, use_aug=False, registration_confidence=opt.confidence_threshold,
                             compute_dtype=opt.compute_dtype)
        ses = ReconstructionSession(torch.load(_default_encoder_path(opt.encoder_path), map_location="cpu"), frames, opt=so)
        conf = np.full(len(files), np.inf)
        xyz, src, sel = ses.point_cloud(torch.load(opt.network, map_location="cpu"), c2w, conf, ses.focal0, dense=opt.dense_point_cloud,
                                        filter_depth=100, opengl=opt.convention == "opengl")
        f, p = np.divmod(src.astype(np.int64), ses.hw)
        clr = source_colours(rgb, sel[f], p, ses.ow)
    write_point_cloud(opt.output_file, xyz, clr)
    _logger.info(f"Done. Wrote point cloud to: {opt.output_file}")
    return 0

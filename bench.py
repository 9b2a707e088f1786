#!/usr/bin/env python
"""bench.py -- throughput of the ACE Zero hot path on MI355X (contract: see the round prompt / DESIGN.md).

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher's environment: re-executes itself under
                                                                torch.distributed.run with N ranks, one per GPU)

A "step" is one training iteration of the scene-coordinate head (ace_trainer.py:499-679) on 5120 patches per
GPU drawn from an 8M-patch synthetic feature buffer resident in HBM (BASELINE config[1]: full ace_zero.py
mapping on one MI355X, default head).  `value` = patches/s of the whole job.  The same run also times the second
half of the metric, DSAC* registration (images/s, scene coordinates resident in HBM, ace_zero's 32 hypotheses /
16 tries), and reports it under "registration".  With N > 1 the feature buffer is sharded across ranks
(each rank owns buffer_patches/N rows... here: its own 8M/N-row shard), every rank runs 5120 rows per step and the
gradient bucket is exchanged over RCCL once per step (reduce-scatter by layer, sharded AdamW, all-gather of the 16-bit copies; weak
scaling: global batch = 5120 N);
registration frames are sharded across ranks with no collective.

Extra objects on the JSON line: "roofline" for the dominant kernel (rowgemm_kernel, bf16 MFMA), measured live
with HIP events on the launch stream, and "cpu_baseline": the oracle (a port -- the reference cannot run on the
GPU box) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 5120                      # train_ace.py:137
FLOP_PER_PATCH = 12_070_912       # fwd 4 198 400 + wgrad 4 198 400 + dgrad 3 674 112
GEMM_FLOP_PER_LAUNCH_PER_ROW = 2 * 512 * 512
MFMA_PEAK_TFLOPS = 2500.0         # MI355X dense bf16 (MI355X_MICROARCH.md)


CSRC = os.path.join(ROOT, "acezero_amd", "csrc")
STEP_SOURCES = ("head_api.hip", "head_kernels.hip", "head_kernels.h", "gemm_common.h", "pose_kernels.hip", "pose_small.hip", "pose_fused.hip")
RANSAC_SOURCES = ("ransac_api.hip", "ransac_math.h", "det_math.h")
WINDOWS = 5                       # timed windows of --steps steps each: `value` = mean over all of them, ms_per_step = their median


def src_digest(files):
    """sha256 over the named kernel sources: stored counter profiles carry it, and a stored figure is only quoted on the JSON line
    while the running build still has the sources it was measured on (otherwise the field is null, never stale)."""
    import hashlib
    h = hashlib.sha256()
    for name in files:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    return h.hexdigest()


def product_digest():
    """sha256 over every product source (acezero_amd/csrc/* and acezero_amd/*.py): the gate of stored WHOLE-PIPELINE figures (the 1000-frame
    session), which depend on all of it."""
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.dirname(CSRC)
    for d in (CSRC, pkg):
        for name in sorted(os.listdir(d)):
            if name.endswith((".hip", ".h", ".py")):
                with open(os.path.join(d, name), "rb") as f:
                    h.update(name.encode())
                    h.update(f.read())
    return h.hexdigest()


def stored_session():
    """profiles/<tag>_session_1000_frames.json (tools/reconstruct_synth.py 1000 144 --store: BASELINE configs[1] as a reconstruction -- 1000
    frames, 8 M-row buffer, the reference's iteration caps) if it was measured on the running product sources, else (None, reason)."""
    tf = os.path.join(ROOT, "profiles", PROFILE_TAG + "_session_1000_frames.json")
    if not os.path.exists(tf):
        return None, f"profiles/{PROFILE_TAG}_session_1000_frames.json not found"
    tj = json.load(open(tf))
    if tj.get("source_digest") != product_digest():
        return None, f"profiles/{PROFILE_TAG}_session_1000_frames.json was measured on different product sources than this build: not quoted"
    return tj, None


PROFILE_TAG = "r06"              # profiles/<tag>_* written by tools/prof_r06.sh: the stored counter / trace pass bench.py may quote
# algorithmic HBM bytes of one 5120-patch step = what must move at least once. Batch rows in (features bf16 1024 B +
# target 8 B + view index 4 B = 1036 B per patch in this layout; the reference's per-patch replicated layout is 1230 B) + the optimiser's
# state traffic: fp32 masters, m, v read and written (24 B / parameter), + two 16-bit compute copies of the wide layers written (4 B).
ALGO_BYTES_PER_PATCH = 1036
ALGO_BYTES_OPTIMISER = 2_103_300 * 24 + 8 * 262_144 * 4
ALGO_BYTES_PER_STEP = ALGO_BYTES_PER_PATCH * BATCH + ALGO_BYTES_OPTIMISER


def parity_table():
    """The tolerances the GPU parity tests assert (tests/golden/parity_tolerances.json: the tests read the same file)."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "parity_tolerances.json")) as f:
            t = json.load(f)
        t.pop("_what", None)
        t["source"] = "tests/golden/parity_tolerances.json (asserted by tests/test_head_gpu.py, tests/test_head_fp16_gpu.py, tests/test_dsac_gpu.py)"
        return t
    except (OSError, ValueError):
        return None


def stored_step_profile():
    """profiles/<tag>_step_hbm_traffic.json if it was measured on the kernel sources of the running build (else (None, reason))."""
    tf = os.path.join(ROOT, "profiles", PROFILE_TAG + "_step_hbm_traffic.json")
    if not os.path.exists(tf):
        return None, f"profiles/{PROFILE_TAG}_step_hbm_traffic.json not found"
    tj = json.load(open(tf))
    if tj.get("source_digest") != src_digest(STEP_SOURCES):
        return None, f"profiles/{PROFILE_TAG}_step_hbm_traffic.json was measured on different kernel sources than this build: not quoted"
    return tj, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--buffer-patches", type=int, default=8_000_000)   # train_ace.py:122
    ap.add_argument("--reg-frames", type=int, default=2048)
    ap.add_argument("--e2e-frames", type=int, default=1088)   # 8 passes of 136 frames
    ap.add_argument("--pose-refinement", default="none", choices=["none", "mlp"])   # ace_zero.py:86 maps every non-seed iteration with mlp
    ap.add_argument("--session-frames", type=int, default=120, help="frames of the in-process ACE0 reconstruction leg (N = 1 only; 0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=("bf16", "fp16"), help="with --headline-only: operand type of the timed step (diagnostics; the default line is bf16 + a dtype_fp16 leg)")
    ap.add_argument("--headline-only", action="store_true", help="only the timed training steps (the leg rocprofv3 is pointed at: tools/prof_r02.sh)")
    # control-flow smoke of the N > 1 path on a ONE-GPU box: every rank on cuda:0, gloo instead of RCCL (not a measurement)
    ap.add_argument("--smoke-same-device", action="store_true")
    # internal: the one-rank RCCL legs as a child process of the default run (dp_world1_legs)
    ap.add_argument("--dp-world1-legs", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def make_buffer(n_patches, device, seed, n_images=1000, grid=(80, 60), feature_dtype=torch.bfloat16):
    """Synthetic training buffer directly in HBM (geometry from acezero_amd.synth, features generated on device).
    grid = feature-map size (w, h): 80 x 60 for 480x640 frames, 93 x 60 for the 480x741 frames of the garden-like leg."""
    from acezero_amd import synth
    views = 2 * n_images
    prob = synth.make_training_problem(seed=seed, n_images=n_images, views_per_image=2, patches_per_view=8, width=8 * grid[0], height=8 * grid[1])
    g = torch.Generator(device=device).manual_seed(seed)
    feats = torch.empty(n_patches, 512, dtype=feature_dtype, device=device)
    chunk = 1 << 20
    for lo in range(0, n_patches, chunk):
        hi = min(n_patches, lo + chunk)
        feats[lo:hi] = torch.randn(hi - lo, 512, generator=g, device=device, dtype=torch.float32).to(feature_dtype)
    view_idx = torch.randint(0, views, (n_patches,), generator=g, device=device, dtype=torch.int32)
    gx = torch.randint(0, grid[0], (n_patches,), generator=g, device=device)
    gy = torch.randint(0, grid[1], (n_patches,), generator=g, device=device)
    target_px = torch.stack([8.0 * (gx + 0.5), 8.0 * (gy + 0.5)], dim=1).float()
    return prob, feats, target_px, view_idx


def bench_training(args, rank, world, device, pose_refinement=None, steps=None, buffer_patches=None, strong=False, windows=WINDOWS,
                   n_images=1000, grid=(80, 60), dtype="bf16"):
    """pose_refinement None -> args.pose_refinement (the headline leg); 'mlp' -> ace_zero's non-seed mapping iterations
    (--pose_refinement mlp --refine_calibration True, ace_zero.py:86,97,262-264).
    strong=True: the REFERENCE's step on N GPUs -- the global batch stays 5120, every rank draws the same permutation of the global
    buffer and runs the rows of each slice that live in its shard (about 5120 / N), one all-reduce of the gradient bucket per step
    (what `torchrun ace_zero.py` runs inside a mapping round); strong=False: 5120 rows per GPU, global batch 5120 N (weak)."""
    pose_refinement = pose_refinement or args.pose_refinement
    steps = steps or args.steps
    buffer_patches = buffer_patches or args.buffer_patches
    from acezero_amd import synth
    from acezero_amd.head import HeadTrainer
    per_rank = buffer_patches // world
    prob, feats, target_px, view_idx = make_buffer(per_rank, device, 2089 + rank, n_images=n_images, grid=grid,
                                                   feature_dtype=torch.float16 if dtype == "fp16" else torch.bfloat16)
    total_iters = windows * steps + args.warmup + 64
    tr = HeadTrainer(prob["mean"], max_batch=BATCH, global_batch=BATCH if strong else BATCH * world, loss_type="tanh", schedule="1cyclepoly",
                     iterations=max(total_iters, 25000), lr_min=0.0005, lr_max=0.003, warmup_iterations=1000, warmup_lr=0.0005,
                     cooldown_iterations=5000, pose_refinement=pose_refinement, dtype=dtype,
                     refine_calibration=pose_refinement != "none", focal_init=float(prob["focal"]))   # ace_zero.py:105-123 mapping settings
    tr.load_flat(torch.from_numpy(synth.init_head_params(1)))
    tr.set_buffer(feats, target_px, view_idx, prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"], prob["view_image"],
                  prob["image_pose_inv"])
    dist = torch.distributed if world > 1 else None
    if strong and world > 1:
        from acezero_amd.parallel import epoch_local_batches
        g = torch.Generator(device=device).manual_seed(8191)                      # the same permutation on every rank
        perm = torch.randperm(per_rank * world, generator=g, device=device)
        local, offs = epoch_local_batches(perm, BATCH, rank * per_rank, (rank + 1) * per_rank)
        nb = len(offs) - 1
        batches = [local[offs[i]:offs[i + 1]] for i in range(min(nb, total_iters))]
    else:
        g = torch.Generator(device=device).manual_seed(8191 + rank)
        perm = torch.randperm(per_rank, generator=g, device=device)               # ace_trainer.py:466
        nb = per_rank // BATCH
        batches = [perm[i * BATCH:(i + 1) * BATCH].contiguous() for i in range(min(nb, total_iters))]

    dp = None
    if dist is not None:
        # the step's exchange over RCCL (acezero_amd/parallel.py): ONE synchronous all-reduce of the flat gradient bucket, AdamW replicated
        # (ACEZ_DP_MODE=sharded: reduce-scatter of the weight gradients by layer, AdamW on the rank's own layers, all-gather of the 16-bit
        # compute copies, one small all-reduce -- parallel.make_data_parallel says why that is not the default)
        from acezero_amd.parallel import make_data_parallel
        dp = make_data_parallel(tr)

    def step(i):
        idx = batches[i % len(batches)]
        if dp is None:
            tr.step(idx, batches[(i + 1) % len(batches)])   # the next batch is known (run_epoch): gathered inside this step's optimiser launch
        else:
            dp.step(idx, batches[(i + 1) % len(batches)])   # the rank's rows of the next batch: gathered inside this step's optimiser launch

    for i in range(args.warmup):
        step(i)
    # `windows` timed windows of exactly `steps` steps, each bracketed by barrier + synchronize (the first one is the contract's
    # timed region; a 20-step window lasts 3 ms, so one window alone moves by several percent from box to box and run to run)
    wins = []
    for w in range(windows):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(args.warmup + w * steps + i)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        wins.append(time.perf_counter() - t0)
    dt = sum(wins) / windows          # mean time of a window of `steps` steps
    st = tr.state()
    assert st["iteration"] == args.warmup + windows * steps and not st["nan"], st
    st["window_ms_per_step"] = [w / steps * 1e3 for w in wins]

    # roofline leg: per-kernel-class durations from HIP events on the launch stream (outside the timed region:
    # the events themselves perturb the step time)
    # every rank runs these 20 steps (with N > 1 a step contains an all-reduce: a rank-0-only loop would deadlock)
    tr.set_profiling(True)
    for i in range(20):
        step(args.warmup + windows * steps + i)
    torch.cuda.synchronize()
    prof = tr.get_profile()
    tr.set_profiling(False)
    if dist is not None:
        dist.barrier()
    return dt, st, prof


def bench_dp_rank_proxy(args, device, rows, mode=None, proxy_world=8, steps=100, rccl_world1=False):
    """SURVEY 8(e), one-GPU proxy of a data-parallel rank's COMPUTE: the launch flow of parallel.make_data_parallel(mode) -- backward on `rows`
    rows of a 5120-row global batch, [the exchange], the update -- with every collective skipped. rows = 5120: a weak-scaling rank;
    rows = 640: a rank of the reference's step split eight ways. ms per step.
      mode "allreduce" (the default): backward + update; the skipped all-reduce leaves the rank's own gradient in the bucket.
      mode "sharded": ShardedDataParallel(proxy_world): staging copies, AdamW on 1 of proxy_world layers, 16-bit export / import.
    rccl_world1: the same flow in a ONE-rank RCCL group with the collectives CALLED: every collective is the identity, so what it adds is
    what the RCCL launches and their stream hand-overs cost per step on this stack, no wire time."""
    from acezero_amd import synth
    from acezero_amd.head import HeadTrainer
    from acezero_amd.parallel import DataParallelTrainer, ShardedDataParallel, dp_mode
    mode = dp_mode(mode)
    n = min(args.buffer_patches, 1_000_000)
    prob, feats, target_px, view_idx = make_buffer(n, device, 2089)
    tr = HeadTrainer(prob["mean"], max_batch=BATCH, global_batch=BATCH, loss_type="tanh", schedule="1cyclepoly", iterations=25000, lr_min=0.0005,
                     lr_max=0.003, warmup_iterations=1000, warmup_lr=0.0005, cooldown_iterations=5000, dtype="bf16")
    tr.load_flat(torch.from_numpy(synth.init_head_params(1)))
    tr.set_buffer(feats, target_px, view_idx, prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"], prob["view_image"], prob["image_pose_inv"])
    if rccl_world1:
        if not torch.distributed.is_initialized():
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        assert torch.distributed.get_backend() == "nccl"
        dp = DataParallelTrainer(tr, force_exchange=True) if mode == "allreduce" else ShardedDataParallel(tr, force_exchange=True)
    else:
        dp = DataParallelTrainer(tr) if mode == "allreduce" else ShardedDataParallel(tr, proxy_world=proxy_world)
        assert not (mode == "allreduce" and dp.exchange)
    perm = torch.randperm(n, generator=torch.Generator(device=device).manual_seed(8191), device=device)
    batches = [perm[i * rows:(i + 1) * rows].contiguous() for i in range(min(n // rows, steps + 20))]
    for i in range(20):
        dp.step(batches[i % len(batches)], batches[(i + 1) % len(batches)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        dp.step(batches[(20 + i) % len(batches)], batches[(21 + i) % len(batches)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = tr.state()
    assert st["iteration"] == steps + 20 and not st["nan"], st   # (the stand-in gradients are finite: every step ran)
    tr.close()
    return dt * 1e3


def bench_registration(args, rank, world, device, h=60, w=80, frames=None):
    from acezero_amd import dsacstar, synth
    n = (frames or args.reg_frames) // world
    base = synth.make_registration_frames(seed=1305 + rank, n_frames=64, h=h, w=w)
    sc = torch.from_numpy(base["scene_coords"]).to(device)
    reps = (n + 63) // 64
    sc = sc.repeat(reps, 1, 1, 1)[:n].contiguous()
    sc += 0.001 * torch.randn(sc.shape, device=device, generator=torch.Generator(device=device).manual_seed(rank))
    intr = [(base["focal"], base["ppx"], base["ppy"])] * n
    prm = dict(hyps=32, thr=10.0, alpha=100.0, max_reproj=100.0, sub=8, max_tries=16)   # ace_zero.py:140-142,233
    ids = [rank * n + i for i in range(n)]
    for _ in range(3):                                     # warm-up: context, code object load; the first launches after the
        dsacstar.register_batch(sc, intr, prm, 1305, ids)  # context's allocations are 2-3x slower than the steady state
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        poses, inl, _ = dsacstar.register_batch(sc, intr, prm, 1305, ids, want_masks=True)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ok = float((inl > 1000).float().mean())
    return n, dt, ok


def ransac_roofline(images_per_s):
    """ransac_kernel is fp64-VALU bound (57.6 KB in, 5 KB out per frame: no HBM or MFMA roof applies). Counter-based: the vector
    instructions per frame come from the committed PMC pass (profiles/r02_ransac_pmc.json, SQ_INSTS_VALU / frames of the same
    2048-frame launch), the rate from the live run; peak = 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz fp64 lane-operations
    (= the 78.6 TFLOP/s fp64 vector peak / 2 flops per FMA)."""
    path = next((q for q in (os.path.join(ROOT, "profiles", tag + "_ransac_pmc.json") for tag in (PROFILE_TAG, "r02")) if os.path.exists(q)),
                os.path.join(ROOT, "profiles", "r02_ransac_pmc.json"))
    pname = "profiles/" + os.path.basename(path)
    peak = 256 * 4 * 16 * 2.4e9 / 1e12
    out = {"bound": "valu_fp64", "kernel": "ransac_kernel<false> (one 256-thread workgroup per 60x80 frame)", "peak": peak,
           "unit": "T lane-ops/s", "achieved": None, "frac": None, "traffic": None}
    # SURVEY section 8(d)'s algorithmic figures beside the counter-based one: scoring = hypotheses x cells x ~32 flop (3x3 mat-vec + t,
    # divide, 2 FMA, norm, clamp, sigmoid) = 4.9 MFLOP per 60x80 frame at 32 hypotheses (the kernel evaluates it in fp64: determinism);
    # the refinement rounds are data dependent and not counted. Bytes: 57.6 KB of coordinates in, 64 B pose + 4.8 KB mask out per frame.
    score_flop = 32 * 4800 * 32
    out["algorithmic_frac"] = images_per_s * score_flop / 1e12 / (2 * peak)   # the R headline figure: SURVEY 8(d)'s scoring flops over the fp64 VALU peak
    out["algorithmic"] = {"scoring_flop_per_frame": score_flop, "scoring_tflops": images_per_s * score_flop / 1e12,
                          "scoring_frac_of_fp64_valu_peak": images_per_s * score_flop / 1e12 / (2 * peak),
                          "hbm_bytes_per_frame": 57600 + 64 + 4800, "hbm_GBps": images_per_s * (57600 + 64 + 4800) / 1e9,
                          "hbm_frac_of_8TBps": images_per_s * (57600 + 64 + 4800) / 8e12,
                          "note": "the algorithmic scoring flops are ~9 % of the lane operations the kernel issues: sampling (P3P per hypothesis and try), the "
                                  "soft-inlier refinement rounds (Gauss-Newton on up to 4800 inliers) and the reductions are the rest; hence the "
                                  "counter-based `frac` above is the utilisation figure"}
    try:
        side = json.load(open(path.replace(".json", ".digest.json")))
        if side.get("source_digest") != src_digest(RANSAC_SOURCES):   # the kernel has changed since the counter pass: nothing stale is quoted
            out["stored_profile"] = pname + " does not match the running build's ransac sources: not used"
            return out
        c = json.load(open(path))["ransac_kernel"]
        insts_per_frame = c["SQ_INSTS_VALU"]["mean_per_launch"] / 2048
        out.update(achieved=images_per_s * insts_per_frame * 64 / 1e12, valu_insts_per_frame=insts_per_frame,
                   lane_utilisation=c["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"] / (c["SQ_INSTS_VALU"]["mean_per_launch"] * 64),
                   stored_profile={"file": pname, "source_digest": side["source_digest"],
                                   "what": "SQ_INSTS_VALU / frames of one 2048-frame launch (a STORED instruction count, checked against the "
                                           "running build's ransac sources); `achieved` = that count x the LIVE images/s of this run"})
        out["frac"] = out["achieved"] / peak
        out["frac_is"] = "issue-slot occupancy (vector instructions issued / fp64 issue peak), NOT an algorithmic roofline: see algorithmic_frac"
    except (OSError, KeyError, ValueError):
        pass
    return out


ENC_FLOP_PER_FRAME = 58.37e9    # 480x640: sum over the 11 convolutions of 2*Ho*Wo*Cin*Cout*k*k (ace_network.py:26-40)
HEAD_FLOP_PER_FRAME = 4800 * (8 * 2 * 512 * 512 + 2 * 512 * 4)


def bench_pipeline(args, rank, world, device, dtype="bf16", legs="all"):
    """(a) images -> encoder -> head -> RANSAC with nothing leaving HBM, (b) training-buffer creation
    (encoder + mask-weighted sampling). 480x640 synthetic grey frames, random-init encoder/head weights (no checkpoints here)."""
    import numpy as np
    from acezero_amd import synth
    from acezero_amd.buffer import BufferBuilder
    from acezero_amd.network import Regressor
    # frames per encoder pass: 128 (round 5; 64 before: a 3x3 layer of 64 frames is 2400 tiles of 118 us = 9.4 waves over 256 CUs, and the
    # last, 37 % full wave costs a whole tile time -- 6.5 % of the launch; at 128 frames it is 1.3 %. tools/e2e_chunk_sweep.py: 64 / 128 /
    # 256 frames -> 0.0646 / 0.0621 / 0.0618 ms per frame; ACEZ_E2E_CHUNK for the comparison)
    # Round 6: 136. A 480 x 640 frame is 18.75 tiles of 256 feature-map pixels, so 136 frames are 2550 tiles = 9.96 waves over 256 CUs for the
    # 256-channel layers (128 frames: 2400 = 9.375 -> ten waves, the last 37 % full) and 19.92 for the 512-channel ones; 137 would be 10.03.
    # tools/bench_encoder.py 128 / 136 on one box: 1091 / 1110 TFLOP/s.
    chunk, total = int(os.environ.get("ACEZ_E2E_CHUNK", "136")), args.e2e_frames // world
    esd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights(seed=4099).items()}
    hsd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict(synth.init_head_params(3)).items()}
    net = Regressor.create_from_split_state_dict(esd, hsd, max_frames=chunk, max_h=480, max_w=640, dtype=dtype)
    img = torch.from_numpy(synth.make_gray_images(seed=1 + rank, n=4, h=480, w=640)).to(device).repeat(chunk // 4, 1, 1, 1).contiguous()
    prm = dict(hyps=32, thr=10.0, alpha=100.0, max_reproj=100.0, sub=8, max_tries=16)
    intr = [(525.0, 320.0, 240.0)] * chunk
    net.register(img, intr, prm, 1305, list(range(chunk)))
    torch.cuda.synchronize()
    # (1) the encoder alone, for its roofline line
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rows = net.encoder.features_rows(img)
    e0.record()
    for c in range(4):
        net.encoder.features_rows(img, out=rows)
    e1.record()
    torch.cuda.synchronize()
    enc_ms_per_frame = e0.elapsed_time(e1) / (4 * chunk)
    # (2) the whole pipeline on `total` frames (the same 32 views repeated: content does not change the work)
    nfr = max(1, total // chunk) * chunk          # whole passes; at least one per rank (a small --e2e-frames over many ranks)
    big = img.repeat(nfr // chunk, 1, 1, 1)
    intr_all = intr * (nfr // chunk)
    net.register(big[:2 * chunk], intr_all[:2 * chunk], prm, 1305)   # warm-up of the grouped path
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    poses, inl, _ = net.register(big, intr_all, prm, 1305)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    enc_ms = enc_ms_per_frame * nfr
    del big
    if legs == "e2e":   # (the fp16 leg: images -> poses and the encoder alone)
        return {"frames": nfr, "e2e_s": dt, "encoder_ms": enc_ms}
    # buffer creation: 1024 samples per view (train_ace.py:128)
    bld = BufferBuilder(net.encoder, capacity=nfr * 1024, samples_per_image=1024, seed=2089)
    eye = torch.eye(4).repeat(chunk, 1, 1)
    K = torch.tensor([[525.0, 0, 320.0], [0, 525.0, 240.0], [0, 0, 1.0]]).repeat(chunk, 1, 1)
    Ki = torch.linalg.inv(K)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for c in range(nfr // chunk):
        bld.add_views(img, None, eye, eye, K, Ki, list(range(chunk)))
    torch.cuda.synchronize()
    dtb = time.perf_counter() - t1
    # point-cloud extraction filter on scene-coordinate maps resident in HBM: 60 x 80 maps of the synthetic room,
    # budgets of a 1000-frame mapping sequence (100 .. 1000 points per frame), launched 256 frames at a time
    from acezero_amd import pointcloud
    fr = synth.make_registration_frames(seed=9 + rank, n_frames=64, noise_sigma=0.002, outlier_ratio=0.2)
    maps = torch.from_numpy(fr["scene_coords"]).to(device).repeat(4, 1, 1, 1).contiguous()
    pinv = torch.from_numpy(np.linalg.inv(fr["poses"]).astype(np.float32)).to(device).repeat(4, 1, 1)
    Kc = torch.tensor([[fr["focal"], 0, fr["ppx"]], [0, fr["focal"], fr["ppy"]], [0, 0, 1.0]], device=device).repeat(256, 1, 1)
    pointcloud.filter_scene_coordinates(maps, pinv, Kc, 100.0, False, 1000)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    npts = 0
    for c in range(8):
        xyz, _, _, _ = pointcloud.filter_scene_coordinates(maps, pinv, Kc, 100.0, False, 1000, first_frame_id=256 * c)
        npts += int(xyz.shape[0])
    torch.cuda.synchronize()
    dtc = time.perf_counter() - t2
    return {"frames": nfr, "e2e_s": dt, "encoder_ms": enc_ms, "buffer_s": dtb, "buffer_rows": bld.n, "cloud_frames": 8 * 256, "cloud_s": dtc,
            "cloud_points": npts}


def bench_session(args, device):
    """the whole ace_zero.py loop in one process on a synthetic, view-consistent sequence (480x640 frames of
    a textured room, 0.5 degrees apart; stand-in encoder weights; reduced iteration counts and augmentation range -- see DESIGN 4c)."""
    from acezero_amd import synth
    from acezero_amd.session import ReconstructionSession, default_options
    n, it = args.session_frames, 4000
    seq = synth.render_room_sequence(seed=2089, n_frames=n, arc_deg=0.5 * n, device=str(device))
    esd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights_bandpass(seed=4099).items()}
    opt = default_options(use_external_focal_length=seq["focal"], try_seeds=2, seed_iterations=it, iterations=it, refit_iterations=it,
                          iterations_max=8, final_refit_posewait=it // 5, learning_rate_warmup_iterations=it // 5, cooldown_iterations=it // 5,
                          aug_rotation=2, aug_scale=1.06, aug_black_white=0.02)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ses = ReconstructionSession(esd, seq["images"], opt=opt, depth=seq["depth"])
    res = ses.reconstruct()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hist = res["history"]
    return {"metric": "in-process ACE0 reconstruction (seed trials + map/register rounds + final refit), wall-clock seconds", "value": dt,
            "unit": "s", "higher_is_better": False, "frames": n, "rounds": len(hist), "registration_rates": [round(h["registration_rate"], 3) for h in hist],
            "training_iterations": int(sum(h.get("iterations", 0) for h in hist) + 2 * it), "max_iterations_per_round": it,
            "note": "synthetic textured room, stand-in encoder (zero-mean smoothed random filters), augmentation +-2 deg / x1.06, 2 seed trials"}


def device_state(device):
    """Clocks / power / partition mode of the GPU while it is busy (rocm-smi beside a ~1.5 s matrix-multiply loop; outside every timed
    region). The pool's boxes differ by 25 % on the same binary (headline step 152-156 us on most, 190-200 us on some): this records
    which kind of box a line came from. Best effort: {} when rocm-smi is missing or its output is not understood."""
    import subprocess
    out = {}
    try:
        a = torch.randn(4096, 4096, device=device, dtype=torch.bfloat16)
        p = subprocess.Popen(["rocm-smi", "-d", str(device.index or 0), "--showclocks", "--showpower", "--showperflevel", "--showcomputepartition",
                              "--showmemorypartition", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.5 and p.poll() is None:
            for _ in range(20):
                a = (a @ a).clamp_(-1, 1)
            torch.cuda.synchronize()
        txt, _ = p.communicate(timeout=20)
        js = json.loads(txt[txt.index("{"):])
        card = next(iter(js.values()))
        for k, v in card.items():
            kl = k.lower()
            if any(w in kl for w in ("sclk", "mclk", "fclk", "socclk", "power", "performance level", "partition")):
                out[k] = v
    except Exception as e:   # noqa: BLE001 (diagnostics only)
        out["error"] = repr(e)[:200]
    return out


def cpu_baseline():
    """Oracle timed on the host cores (kind "port": the reference itself is not on the GPU box)."""
    from acezero_amd import synth
    from concurrent.futures import ThreadPoolExecutor
    from oracle import dsac_oracle, head_oracle
    cores = os.cpu_count() or 1
    tcores = min(cores, 32)   # torch-CPU GEMMs of this size slow down badly when oversubscribed (256 threads: 16 s/step)
    torch.set_num_threads(tcores)
    prob = synth.make_training_problem(seed=3, n_images=20, views_per_image=2, patches_per_view=128)
    cfg = dict(loss_type="tanh", schedule="1cyclepoly", iterations=25000, lr_min=0.0005, lr_max=0.003, warmup_iterations=1000,
               warmup_lr=0.0005, cooldown_iterations=5000, cooldown_trigger_percent=0.7, global_batch=BATCH, soft_clamp=50.0,
               soft_clamp_min=1.0, hard_clamp=1000.0, depth_min=0.1, depth_max=1000.0, depth_target=10.0, inlier_px_threshold=10.0)
    orc = head_oracle.TrainerOracle(head_oracle.init_params(1), prob["mean"], cfg, mode="fp32")
    rng = np.random.default_rng(0)
    idx = rng.integers(0, prob["features"].shape[0], BATCH)
    pp = synth.expand_per_patch(prob, idx)
    b = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in pp.items()}
    orc.step(b["features"], b)
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < 12.0 and steps < 120:
        orc.step(b["features"], b)
        steps += 1
    t_train = (time.perf_counter() - t0) / max(steps, 1)
    kind, train_what = "port", f"{steps} steps of 5120 patches, oracle/head_oracle.py fp32 on torch-CPU with {tcores} of {cores} host threads"
    if os.path.isdir("/root/reference"):
        # the reference itself (build container only: /root/reference does not exist on the GPU box): the unmodified
        # TrainerACE.training_step through the stub-import recipe of tests/golden/make_head_golden.py
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("make_head_golden", os.path.join(ROOT, "tests", "golden", "make_head_golden.py"))
            mk = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mk)
            t_ref, n_ref = mk.time_reference_step(batch=BATCH, budget_s=12.0, max_steps=40, threads=tcores)
            t_train, kind = t_ref, "reference"
            train_what = (f"{n_ref} calls of the reference's TrainerACE.training_step (ace_trainer.py:499-679) at 5120 patches, CPU PyTorch fp32, "
                          f"{tcores} of {cores} host threads (the reference pins them to 1 at import, ace_trainer.py:5-8: overridden)")
        except Exception as e:   # noqa: BLE001 -- a broken reference import must not take the bench down
            train_what += f" (reference import failed: {type(e).__name__})"
    fr = synth.make_registration_frames(seed=5, n_frames=16)
    dsac_oracle.lib()

    def one(i):
        return dsac_oracle.forward_rgb(fr["scene_coords"][i % 16], 32, 10.0, fr["focal"], fr["ppx"], fr["ppy"], 100.0, 100.0, 8, 1305, i, 16)["inliers"]
    nfr = 16 * cores
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(one, range(nfr)))
    t_reg = time.perf_counter() - t0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(min(12, cores)) as ex:   # register_mapping.py:6-9 runs dsacstar with OMP_NUM_THREADS = 12
        list(ex.map(one, range(16 * min(12, cores))))
    t_reg12 = (time.perf_counter() - t0) / (16 * min(12, cores))
    # encoder (the end-to-end registration leg's dominant cost): torch conv2d fp32 on the host cores, 2 frames of 480 x 640
    from oracle import encoder_oracle
    enc = encoder_oracle.EncoderOracle(encoder_oracle.init_weights(seed=4099), "fp32")
    img = torch.from_numpy(synth.make_gray_images(seed=1, n=2, h=480, w=640))
    with torch.no_grad():
        enc.forward(img[:1])
        t0 = time.perf_counter()
        enc.forward(img)
        t_enc = (time.perf_counter() - t0) / 2
    from oracle import cloud_oracle
    frc = synth.make_registration_frames(seed=9, n_frames=8, noise_sigma=0.002, outlier_ratio=0.2)
    pinv = np.linalg.inv(frc["poses"]).astype(np.float32)[:, :3]
    Kc = np.array([[frc["focal"], 0, frc["ppx"]], [0, frc["focal"], frc["ppy"]], [0, 0, 1]], np.float32)
    t0 = time.perf_counter()
    for rep in range(4):
        cloud_oracle.point_cloud(frc["scene_coords"], pinv, [Kc] * 8, 100.0, False, 1000)
    t_cloud = (time.perf_counter() - t0) / 32
    ref_stored = None
    try:
        with open(os.path.join(ROOT, "profiles", "r04_cpu_reference_training_step.json")) as f:
            rs = json.load(f)
        ref_stored = {"kind": "reference (stored)", "value": rs["patches_per_s"], "unit": "patches/s", "cores": rs["threads"], "cpu": rs["cpu"],
                      "seconds_per_step": rs["seconds_per_step_median"], "file": "profiles/r04_cpu_reference_training_step.json",
                      "what": "the reference's own TrainerACE.training_step timed in the build container (the GPU box has no reference checkout); not this box's "
                              "cores -- the live figure beside it is"}
    except (OSError, ValueError, KeyError):
        pass
    return {"value": BATCH / t_train, "unit": "patches/s", "cores": tcores, "kind": kind, "reference_stored": ref_stored,
            "kind_note": "reference = the reference's own training_step (only where /root/reference exists); port = oracle/ restatement "
                         "(the GPU box has no reference checkout). The registration figures are always the oracle port: the reference's "
                         "dsacstar needs OpenCV 4.4.0, which is not available.",
            "registration_images_per_s_12_threads": 1.0 / t_reg12,
            "point_cloud_frames_per_s": 1.0 / t_cloud,
            "sample": train_what + f"; registration: {nfr} frames, oracle/dsac_oracle.cpp, {cores} threads (and with the reference's 12)",
            "registration_images_per_s": nfr / t_reg,
            "encoder_images_per_s": 1.0 / t_enc,
            "encoder_sample": f"2 frames of 480x640, oracle/encoder_oracle.py (torch conv2d fp32), {tcores} threads"}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: re-execute under torch.distributed.run with N ranks on this node, one per GPU
    (the command the driver uses; 127.0.0.1 rendezvous on a free port). Never returns. A box with fewer than N GPUs fails loudly instead
    of printing an n_gpus = 1 line (--smoke-same-device: all ranks on cuda:0 over gloo, a control-flow check on a one-GPU box)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if not args.smoke_same_device and have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node; refusing to run fewer ranks than asked for")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def dp_world1_legs(args):
    """What the RCCL calls themselves cost per step (one rank: no wire time), against the same flow with the calls skipped: the legs that
    form a one-rank RCCL group run in a CHILD process under a hard time limit -- a box on which RCCL cannot bootstrap (or hangs doing so)
    loses these auxiliary figures, never the line. Returns (figures or None, error or None)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--dp-world1-legs", "--buffer-patches", str(args.buffer_patches)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return None, ("rc %d: " % r.returncode) + r.stderr[-300:]
        return json.loads(lines[-1]), None
    except Exception as e:
        return None, repr(e)[:300]


def dp_world1_child(args):
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    line_out = claim_stdout()
    out = {"allreduce_rccl": bench_dp_rank_proxy(args, device, 5120, rccl_world1=True),
           "sharded_skipped": bench_dp_rank_proxy(args, device, 5120, mode="sharded", proxy_world=1),
           "sharded_rccl": bench_dp_rank_proxy(args, device, 5120, mode="sharded", rccl_world1=True),
           "sharded_rank_of_8": bench_dp_rank_proxy(args, device, 5120, mode="sharded")}
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    print(json.dumps(out), file=line_out, flush=True)


def dp_mode_name():
    from acezero_amd.parallel import dp_mode
    return dp_mode()


def claim_stdout():
    """The contract is ONE JSON line on stdout, and libraries write to file descriptor 1 behind Python's back (RCCL prints a five-line
    version banner there at the first communicator of a process -- found when the one-rank RCCL leg joined the default run). The real
    stdout is kept for the line; descriptor 1 -- Python's sys.stdout and every C library's -- goes to stderr from here on. A caller that
    has replaced sys.stdout by an object of its own (tests) keeps it."""
    try:
        if sys.stdout.fileno() != 1:
            return sys.stdout
    except (AttributeError, OSError, ValueError):
        return sys.stdout
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(line_fd, "w")


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    if args.dp_world1_legs:
        return dp_world1_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    line_out = claim_stdout()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if env_world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={env_world}: the launcher's rank count and --gpus must agree")
    if args.smoke_same_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    rank, world, rccl_ranks = 0, 1, 0
    if env_world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.smoke_same_device:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=device)
        # the line's n_gpus is what the initialised process group says, not what a flag or an environment variable claims
        rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        rccl_ranks = world if torch.distributed.get_backend() == "nccl" else 0
    assert world == args.gpus, (world, args.gpus)

    dt, st, prof = bench_training(args, rank, world, device, dtype=args.dtype if args.headline_only else "bf16")
    dev_state = device_state(device) if rank == 0 else None
    if args.headline_only:
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t[0])
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "ACE patches/sec", "value": BATCH * world * args.steps / dt, "unit": "patches/s", "n_gpus": world,
                              "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(np.median(st["window_ms_per_step"])),
                              "ms_per_step_mean": dt / args.steps * 1e3, "window_ms_per_step": st["window_ms_per_step"], "headline_only": True,
                              "pose_refinement": args.pose_refinement, "dtype": args.dtype,
                              "per_class_us_per_step": {k: v[0] / 20 * 1e3 for k, v in prof.items()}, "final_loss": st["loss"], "device_state": dev_state}),
                  file=line_out, flush=True)
        return
    # the headline step with fp16 operands (the reference's autocast precision; HeadTrainer(dtype="fp16")): same kernels, same rate
    dt_f16, st_f16, _ = bench_training(args, rank, world, device, steps=100, buffer_patches=min(args.buffer_patches, 2_000_000), dtype="fp16")
    dt_ref, st_ref, _ = bench_training(args, rank, world, device, pose_refinement="mlp", steps=100, buffer_patches=min(args.buffer_patches, 2_000_000))
    # the step every non-seed round of ace_zero.py runs, at the reference's operand precision
    dt_ref16, st_ref16, _ = bench_training(args, rank, world, device, pose_refinement="mlp", steps=100, buffer_patches=min(args.buffer_patches, 2_000_000),
                                           dtype="fp16")
    # BASELINE configs[2] (Mip-NeRF 360 garden-like): 185 frames of 480x741 -> 60x93 maps, focal refinement in the loop
    dt_gar, st_gar, _ = bench_training(args, rank, world, device, pose_refinement="mlp", steps=100, buffer_patches=min(args.buffer_patches, 2_000_000),
                                       n_images=185, grid=(93, 60))
    ngar, dt_gar_reg, gar_ok = bench_registration(args, rank, world, device, h=60, w=93, frames=1024)
    dt_strong = None
    if world > 1:   # the reference's step (global batch 5120) split over the ranks
        dt_strong, st_strong, _ = bench_training(args, rank, world, device, steps=args.steps, buffer_patches=min(args.buffer_patches, 2_000_000), strong=True)
    nreg, dt_reg, reg_ok = bench_registration(args, rank, world, device)
    pipe = bench_pipeline(args, rank, world, device)
    pipe16 = bench_pipeline(args, rank, world, device, dtype="fp16", legs="e2e")
    # one-GPU proxies of a data-parallel rank's compute (N = 1 only: what a rank does between its collectives)
    dp_proxy = {rows: bench_dp_rank_proxy(args, device, rows) for rows in (5120, 640)} if world == 1 else None
    dp_w1, dp_w1_error = (None, None) if world > 1 else dp_world1_legs(args)
    sess = bench_session(args, device) if world == 1 and args.session_frames > 0 else None
    if world > 1:
        t = torch.tensor([dt, dt_reg, dt_ref, dt_strong, dt_gar, dt_gar_reg, dt_f16] + st["window_ms_per_step"], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, dt_reg, dt_ref, dt_strong, dt_gar, dt_gar_reg, dt_f16 = (float(x) for x in t[:7])
        st["window_ms_per_step"] = [float(x) for x in t[7:]]
    if rank == 0:
        patches_per_s = BATCH * world * args.steps / dt
        gemm_ms, gemm_n = 0.0, 0
        for k in ("gemm_fwd", "gemm_dgrad"):
            gemm_ms += prof[k][0]
            gemm_n += prof[k][1]
        avg_s = gemm_ms / max(gemm_n, 1) * 1e-3
        achieved = BATCH * GEMM_FLOP_PER_LAUNCH_PER_ROW / avg_s / 1e12 if avg_s > 0 else 0.0
        seq = os.environ.get("ACEZ_SEQ", "1") != "0"
        gemm_kernel_name = ("rowseq_kernel (the 8 forward / the 7 input-gradient 5120x512x512 bf16 layers of a step as ONE launch each; figures per layer)"
                            if seq else "rowgemm80_kernel (5120x512x512 bf16, fwd + dgrad launches)")
        gemm_note = ("HIP events on the launch stream around the two chain launches of a step; achieved = 15 layers' FLOPs / their summed durations = one "
                     "layer's FLOPs / avg_launch_us (launches_timed counts layers); rocprofv3 kernel durations are in profiles/" + PROFILE_TAG + "_kernel_stats_rocprofv3_headline_only_trace.csv"
                     if seq else
                     "HIP events on the launch stream around each chain of dependent rowgemm launches (8 fwd, 7 dgrad per step): average start-to-start "
                     "cadence incl. the ~1-2 us kernel boundary; rocprofv3 kernel durations are in profiles/")
        # Counter figures cannot be collected inside this run (they need rocprofv3 around the process: tools/prof_r03.sh); the stored
        # pass is quoted only if it was taken on the sources this build was compiled from, and it names the kernel it belongs to.
        traffic, traffic_source, mfma_busy, stored = None, None, None, None
        step_prof = None
        tj, why_not = stored_step_profile()
        if seq and tj is not None:
            ks = {k: v for k, v in tj["kernels"].items() if "rowseq_kernel" in k and "bytes_per_layer" in v}
            if len(ks) == 2:
                layers = sum(v["layers_per_launch"] for v in ks.values())
                traffic = sum(v["bytes_per_launch"] for v in ks.values()) / layers       # per unit (one layer), like `achieved`
                if all("mfma_busy_frac" in v for v in ks.values()):
                    mfma_busy = sum(v["mfma_busy_frac"] * v["avg_ns"] for v in ks.values()) / sum(v["avg_ns"] for v in ks.values())
                traffic_source = (f"profiles/{PROFILE_TAG}_step_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE x fetch_factor + WRITE_SIZE x write_factor (separate "
                                  "passes; the factors are calibrated on this kernel's access pattern with tools/pmc_calib.hip and stored in the file) of "
                                  "rowseq_kernel<false> (8 layers) and rowseq_kernel<true> (7 layers), the kernels timed here, per layer; a STORED "
                                  "measurement taken on the same kernel sources as this build (source_digest matches)")
                stored = {"file": f"profiles/{PROFILE_TAG}_step_hbm_traffic.json", "source_digest": tj["source_digest"], "kernels": sorted(ks),
                          "calibration": tj.get("calibration")}
            # the whole step from the same stored pass: every kernel's duration and counter bytes, so that each `frac` on this line can be
            # recomputed from profiles/ alone
            rows = {k: {"us": v.get("avg_ns", 0.0) / 1e3, "counter_bytes": v.get("bytes_per_launch"), "mfma_busy_frac": v.get("mfma_busy_frac")}
                    for k, v in tj["kernels"].items() if v.get("calls", 0) >= 100}
            cb = sum(r["counter_bytes"] or 0.0 for r in rows.values())
            step_prof = {"file": stored["file"] if stored else None, "per_kernel": rows, "sum_of_kernel_us": sum(r["us"] for r in rows.values()),
                         "counter_bytes_per_step": cb, "algorithmic_bytes_per_step": ALGO_BYTES_PER_STEP,
                         "counter_over_algorithmic": cb / ALGO_BYTES_PER_STEP if cb else None,
                         "algorithmic_note": "SURVEY 8(d): 1036 B per patch x 5120 + optimiser state (24 B per parameter read + written, two 16-bit copies of the wide "
                                             "layers written) = what must cross HBM at least once per step; saved activations (5120 x 512 x 2 B per layer output, "
                                             "gradient and residual tile) round-trip through the 256 MB Infinity Cache and are counted by the L2-side counters"}
        elif seq:
            traffic_source = why_not
        # the single longest kernel of the step (the input-gradient chain, rowseq_kernel<true>) beside the average over both chains
        dg_us = prof["gemm_dgrad"][0] / 20 * 1e3
        dg_layers = prof["gemm_dgrad"][1] / 20
        dg_tflops = BATCH * GEMM_FLOP_PER_LAUNCH_PER_ROW * dg_layers / (dg_us * 1e-6) / 1e12 if dg_us > 0 else 0.0
        dominant = {"kernel": "rowseq_kernel<true> (the 7 input-gradient layers, one launch)" if seq else "the 7 input-gradient rowgemm80 launches",
                    "us_per_step": dg_us, "layers": dg_layers, "achieved": dg_tflops, "frac": dg_tflops / MFMA_PEAK_TFLOPS,
                    "forward_chain": {"us_per_step": prof["gemm_fwd"][0] / 20 * 1e3, "layers": prof["gemm_fwd"][1] / 20,
                                      "frac": (BATCH * GEMM_FLOP_PER_LAUNCH_PER_ROW * (prof["gemm_fwd"][1] / 20) / max(prof["gemm_fwd"][0] / 20 * 1e-3, 1e-12) / 1e12) / MFMA_PEAK_TFLOPS}}
        wg_s = prof["wgrad"][0] / max(prof["wgrad"][1], 1) * 1e-3
        wg_tflops = BATCH * 8 * GEMM_FLOP_PER_LAUNCH_PER_ROW / wg_s / 1e12 if wg_s > 0 else 0.0
        rr = ransac_roofline(nreg * world / dt_reg)
        enc_frac = ENC_FLOP_PER_FRAME * pipe["frames"] / (pipe["encoder_ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS
        # Flat scalars: the driver's record keeps the scalars of `config` / `roofline` and drops nested objects and unknown top-level keys, so
        # the figures a reader needs first are repeated there; `summary` is the same for a reader of the raw line.
        summary = {"dominant_kernel_frac": dominant["frac"], "dominant_kernel_us": dg_us, "forward_chain_frac": dominant["forward_chain"]["frac"],
                   "wgrad_opt_frac": wg_tflops / MFMA_PEAK_TFLOPS, "whole_step_frac": patches_per_s / world * FLOP_PER_PATCH / (MFMA_PEAK_TFLOPS * 1e12),
                   "refinement_ms_per_step": float(np.median(st_ref["window_ms_per_step"])), "fp16_ms_per_step": float(np.median(st_f16["window_ms_per_step"])),
                   "refinement_fp16_ms_per_step": float(np.median(st_ref16["window_ms_per_step"])),
                   "e2e_fp16_images_per_s": pipe16["frames"] * world / pipe16["e2e_s"], "encoder_fp16_ms_per_frame": pipe16["encoder_ms"] / pipe16["frames"],
                   "dp_rank_compute_ms_5120": None if dp_proxy is None else dp_proxy[5120], "dp_rank_compute_ms_640": None if dp_proxy is None else dp_proxy[640],
                   "dp_world1_exchange_rccl_ms": None if dp_w1 is None else dp_w1["allreduce_rccl"],
                   "garden_ms_per_step": float(np.median(st_gar["window_ms_per_step"])),
                   "registration_images_per_s": nreg * world / dt_reg, "registration_e2e_images_per_s": pipe["frames"] * world / pipe["e2e_s"],
                   "encoder_frac_of_mfma_peak": enc_frac, "encoder_ms_per_frame": pipe["encoder_ms"] / pipe["frames"],
                   "buffer_rows_per_s": pipe["buffer_rows"] * world / pipe["buffer_s"],
                   "ransac_algorithmic_frac": rr["algorithmic_frac"], "ransac_issue_occupancy": rr["frac"]}
        # BASELINE configs[1] as a RECONSTRUCTION (1000 frames, 8 M-row buffer, the reference's iteration caps): a stored figure (its run is
        # 25-30 s of GPU on top of a 20 s render), quoted only while the product sources are the ones it was measured on
        ss, ss_why = stored_session()
        tm = (ss or {}).get("timings") or {}
        tot = (ss or {}).get("reconstruction_s")
        summary.update({"session_1000_s": tot, "session_loop_frac": (tm.get("loop_s") / tot) if tot and tm else None,
                        "session_buffer_frac": (tm.get("buffer_s") / tot) if tot and tm else None,
                        "session_register_frac": (tm.get("register_s") / tot) if tot and tm else None,
                        "session_1000_registered": (ss or {}).get("registered"), "session_1000_source": ss_why or "profiles/%s_session_1000_frames.json" % PROFILE_TAG})
        out = {
            "metric": "ACE patches/sec", "value": patches_per_s, "unit": "patches/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": float(np.median(st["window_ms_per_step"])), "ms_per_step_mean": dt / args.steps * 1e3,
            "window_ms_per_step": st["window_ms_per_step"], "windows": WINDOWS, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "7-Scenes-chess-like ace_zero mapping step: 8M-patch bf16 feature buffer in HBM, batch 5120 per GPU, "
                                   "default head (1 block, 2 103 300 params), tanh loss, 1cyclepoly AdamW, pose_refinement " + args.pose_refinement,
                       "buffer_patches": args.buffer_patches, "batch_per_gpu": BATCH, "global_batch": BATCH * world,
                       "parallelism": f"dp{world}" + ("" if world == 1 else " (" + ("one all-reduce of the gradient bucket, replicated AdamW" if dp_mode_name() == "allreduce" else "reduce-scatter of the weight gradients by layer, AdamW on the owned layers, all-gather of the 16-bit weights") + ")")},
            "summary": summary,
            "whole_step_flop_frac_of_mfma_peak": patches_per_s / world * FLOP_PER_PATCH / (MFMA_PEAK_TFLOPS * 1e12),
            "strong_scaling": None if dt_strong is None else {
                "metric": "ACE patches/sec, the reference's step: global batch 5120 split over the ranks by buffer shard, one gradient exchange per step (see `collective`)",
                "value": BATCH * args.steps / dt_strong, "unit": "patches/s", "ms_per_step": dt_strong / args.steps * 1e3, "scaling": "strong",
                "collective": dp_mode_name(),
                "global_batch": BATCH, "rows_per_gpu": BATCH / world},
            "refinement_step": {"metric": "ACE patches/sec with --pose_refinement mlp --refine_calibration True (every non-seed mapping iteration of ace_zero.py)",
                                "value": BATCH * world * 100 / dt_ref, "unit": "patches/s", "ms_per_step": dt_ref / 100 * 1e3, "steps": 100,
                                "ms_per_step_median": float(np.median(st_ref["window_ms_per_step"])), "window_ms_per_step": st_ref["window_ms_per_step"],
                                "n_images": 1000, "final_loss": st_ref["loss"]},
            "dp_rank_proxy": None if dp_proxy is None else {
                "what": "ms per step of ONE data-parallel rank's compute on one GPU, the default exchange (parallel.make_data_parallel: backward on "
                        "`rows` rows of a 5120-row global batch, ONE all-reduce of the gradient bucket, replicated AdamW) with the collective skipped. "
                        "Multi-GPU itself is unmeasured on hardware; DESIGN.md section 7 adds the wire estimate to these",
                "mode": dp_mode_name(), "rows_5120_ms": dp_proxy[5120], "rows_640_ms": dp_proxy[640],
                "world1_exchange": {"error": dp_w1_error} if dp_w1 is None else {
                    "what": "5120 rows, a ONE-rank RCCL group with the collectives CALLED (identity transfers): what the RCCL launches and their "
                            "stream hand-overs cost per step on this stack, without any wire time. allreduce_rccl_ms against rows_5120_ms: the "
                            "default; sharded_*: ACEZ_DP_MODE=sharded as the only rank with its collectives skipped / called, and as rank 0 of 8 "
                            "(AdamW on one layer) with them skipped",
                    "allreduce_rccl_ms": dp_w1["allreduce_rccl"], "sharded_skipped_ms": dp_w1["sharded_skipped"], "sharded_rccl_ms": dp_w1["sharded_rccl"],
                    "sharded_rank_of_8_skipped_ms": dp_w1["sharded_rank_of_8"]}},
            "dtype_fp16": {"metric": "ACE patches/sec, the headline step with fp16 operands (the reference's autocast precision, compute_dtype fp16)",
                           "value": BATCH * world * 100 / dt_f16, "unit": "patches/s", "ms_per_step": dt_f16 / 100 * 1e3,
                           "ms_per_step_median": float(np.median(st_f16["window_ms_per_step"])), "final_loss": st_f16["loss"],
                           "refinement_ms_per_step_median": float(np.median(st_ref16["window_ms_per_step"])),
                           "registration_e2e_images_per_s": pipe16["frames"] * world / pipe16["e2e_s"],
                           "encoder_ms_per_frame": pipe16["encoder_ms"] / pipe16["frames"]},
            "garden_like": {"metric": "BASELINE configs[2] shape: 185 frames of 480x741 (60x93 feature maps), --pose_refinement mlp --refine_calibration True",
                            "training": {"value": BATCH * world * 100 / dt_gar, "unit": "patches/s", "ms_per_step": dt_gar / 100 * 1e3,
                                         "ms_per_step_median": float(np.median(st_gar["window_ms_per_step"])), "n_images": 185,
                                         "focal_scale_after": st_gar["focal_scale"], "final_loss": st_gar["loss"]},
                            "registration": {"value": ngar * world / dt_gar_reg, "unit": "images/s", "frames": ngar * world, "map": "60x93",
                                             "hypotheses": 32, "max_tries": 16, "frac_frames_registered": gar_ok}},
            "registration": {"metric": "DSAC* images-registered/sec", "value": nreg * world / dt_reg, "unit": "images/s",
                             "frames": nreg * world, "hypotheses": 32, "max_tries": 16, "frac_frames_registered": reg_ok,
                             "note": "RANSAC only, 60x80 scene coordinates resident in HBM"},
            "registration_e2e": {"metric": "images registered/sec, grey 480x640 frame -> encoder -> head -> RANSAC, device resident",
                                 "value": pipe["frames"] * world / pipe["e2e_s"], "unit": "images/s", "frames": pipe["frames"] * world,
                                 "encoder_ms_per_frame": pipe["encoder_ms"] / pipe["frames"],
                                 "encoder_tflops": ENC_FLOP_PER_FRAME * pipe["frames"] / (pipe["encoder_ms"] * 1e-3) / 1e12,
                                 "encoder_frac_of_mfma_peak": ENC_FLOP_PER_FRAME * pipe["frames"] / (pipe["encoder_ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                                 "whole_pipeline_tflops": (ENC_FLOP_PER_FRAME + HEAD_FLOP_PER_FRAME) * pipe["frames"] / pipe["e2e_s"] / 1e12,
                                 "note": "rank 0's numbers x world; random-init weights, synthetic frames; encoder = convgemm_kernel (implicit GEMM, bf16 MFMA)"},
            "buffer_creation": {"metric": "training-buffer rows/sec (encoder + 1024 mask-weighted samples per 480x640 view)",
                                "value": pipe["buffer_rows"] * world / pipe["buffer_s"], "unit": "patches/s",
                                "views_per_s": pipe["frames"] * world / pipe["buffer_s"]},
            "ace_zero_session": sess,
            "point_cloud_filter": {"metric": "point-cloud extraction frames/sec (60x80 scene-coordinate maps in HBM -> filtered [N,3] list)",
                                   "value": pipe["cloud_frames"] * world / pipe["cloud_s"], "unit": "frames/s",
                                   "points_per_frame": pipe["cloud_points"] / pipe["cloud_frames"],
                                   "map_bytes_per_s": pipe["cloud_frames"] * world * 57600 / pipe["cloud_s"]},
            "roofline": {"bound": "mfma", "kernel": gemm_kernel_name, "achieved": achieved,
                         "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": traffic,
                         **{"summary_" + k: v for k, v in summary.items()},
                         "dominant_kernel": dominant, "stored_step_profile": step_prof,
                         "traffic_source": traffic_source, "mfma_busy_frac": mfma_busy, "stored_profile": stored,
                         "avg_launch_us": avg_s * 1e6, "launches_timed": gemm_n,
                         "per_class_us_per_step": {k: v[0] / 20 * 1e3 for k, v in prof.items()},
                         "note": gemm_note},
            "roofline_wgrad": {"bound": "mfma", "kernel": "wgrad_opt_kernel (8 layers x 512x512x5120 bf16 weight gradients in one launch; since round 4 the same "
                                                          "launch also exchanges the split-K halves, applies AdamW to the wide layers and the small parameters and "
                                                          "closes the step's schedule: there is no optimiser launch any more)" if "adamw" not in prof or prof["adamw"][0] == 0
                               else "wgrad_kernel (8 layers x 512x512x5120 bf16 in one launch)", "achieved": wg_tflops,
                               "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": wg_tflops / MFMA_PEAK_TFLOPS, "traffic": None,
                               "avg_launch_us": wg_s * 1e6,
                               "note": "achieved = the weight-gradient FLOPs over the WHOLE launch, optimiser epilogue included (round 3: wgrad_kernel alone 27-29 us + "
                                       "a 23-26 us optimiser launch)"},
            "roofline_ransac": rr,
            "parity": parity_table(),
            "precision_note": "dtype bf16 is what BASELINE north_star names; the reference runs fp16 autocast (ace_trainer.py:517-518): `dtype_fp16` is the "
                              "same step at the reference's operand precision (scene coordinates within 2e-3 of the reference's fp32 arithmetic, bf16: 3e-2)",
            "final_loss": st["loss"], "device_state": dev_state,
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), file=line_out, flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

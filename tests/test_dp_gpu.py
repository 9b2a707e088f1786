"""GPU: the data-parallel product path with TWO ranks on ONE GPU over gloo (the driver's multi-GPU run is the only place where N
real GPUs meet; a gpurun box has one). Same code as under `torchrun --nproc-per-node N ace_zero.py ...` with RCCL:
  * HeadTrainer backward -> all_reduce(grad) -> update with the reference's batch composition (parallel.epoch_local_batches):
    the summed gradient equals the single-rank gradient of the same 5120-row batch, replicas stay BIT-identical, and the
    parameters track the single-rank run;
  * ReconstructionSession on two ranks (frames sharded for encoding and registration, buffer sharded by image, seed trials on
    different ranks): both ranks return identical poses, and the mapping round relocalises held-out frames."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers

pytestmark = pytest.mark.gpu
B = 5120


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _collect(procs, q, timeout):
    """One result per worker; fails at once when a worker has died (a crashed rank must not cost the GPU box the full timeout)."""
    import queue
    import time
    out, t0 = [], time.time()
    while len(out) < len(procs):
        try:
            out.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > timeout:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError(f"worker exit codes {[p.exitcode for p in procs]} after {time.time() - t0:.0f}s")
    return out


def _problem():
    from acezero_amd import synth
    prob = synth.make_training_problem(seed=helpers.SEED + 7, n_images=24, views_per_image=2, patches_per_view=512)
    prob["features"] = torch.from_numpy(prob["features"]).to(torch.bfloat16).to(torch.float32).numpy()
    from oracle import head_oracle
    return prob, head_oracle.init_params(helpers.SEED + 1)


def _make_trainer(prob, flat0, lo, hi):
    from acezero_amd.head import HeadTrainer
    tr = HeadTrainer(prob["mean"], max_batch=B, global_batch=B, loss_type="tanh", schedule="1cyclepoly", iterations=100, lr_min=1e-4, lr_max=6e-4,
                     warmup_iterations=10, warmup_lr=1e-4, cooldown_iterations=20, refine_calibration=True, focal_init=float(prob["focal"]))
    tr.load_flat(flat0)
    tr.set_buffer(prob["features"][lo:hi], prob["target_px"][lo:hi], prob["view_idx"][lo:hi], prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"],
                  prob["view_image"], prob["image_pose_inv"])
    return tr


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from acezero_amd import parallel
    prob, flat0 = _problem()
    n = prob["features"].shape[0]
    lo, hi = parallel.shard_range(n, rank, world)
    tr = _make_trainer(prob, flat0, lo, hi)
    gen = torch.Generator(device="cuda").manual_seed(8191)
    perm = torch.randperm(n, generator=gen, device="cuda")               # every rank draws the same permutation
    local, offs = parallel.epoch_local_batches(perm, B, lo, hi)
    grads = []
    for b in range(3):
        tr.backward(local[offs[b]:offs[b + 1]])
        dist.all_reduce(tr.grad)
        grads.append(tr.grad.cpu().numpy().copy())
        tr.update()
    torch.cuda.synchronize()
    q.put((rank, grads, tr.params.cpu().numpy(), tr.state(), perm[:3 * B].cpu().numpy(), [offs[b + 1] - offs[b] for b in range(3)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_equals_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(_collect(procs, q, 300), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, p0, s0, perm, c0), (_, g1, p1, s1, _, c1) = res
    assert all(a + b == B for a, b in zip(c0, c1)), "every batch keeps its 5120 rows over the two shards"
    assert min(c0 + c1) > 2000                                            # (about half each)
    for a, b in zip(g0, g1):
        assert np.array_equal(a, b), "all-reduced buckets are identical on both ranks"
    assert np.array_equal(p0, p1) and s0 == s1, "replicas stay bit-identical without a weight broadcast"
    # the single-rank run of the same three batches
    prob, flat0 = _problem()
    tr = _make_trainer(prob, flat0, 0, prob["features"].shape[0])
    npar = flat0.numel()
    for b in range(3):
        tr.backward(torch.from_numpy(perm[b * B:(b + 1) * B]).cuda())
        g = tr.grad.cpu().numpy()
        rel = np.linalg.norm(g0[b][:npar] - g[:npar]) / np.linalg.norm(g[:npar])
        assert rel < (2e-3 if b == 0 else 2e-2), rel                       # b > 0: the two runs' weights differ in the last bits already
        assert abs(g0[b][npar] - g[npar]) < 1e-3 * abs(g[npar]) and (b > 0 or g0[b][npar + 1] == g[npar + 1])   # loss sum, inlier count
        tr.update()
    d_dp, d_1 = p0 - flat0.numpy(), tr.params.cpu().numpy() - flat0.numpy()
    assert np.linalg.norm(d_dp - d_1) < 0.1 * np.linalg.norm(d_1)
    assert tr.state()["iteration"] == s0["iteration"] == 3 and abs(tr.state()["lr"] - s0["lr"]) < 1e-15


def _session_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from acezero_amd import synth
    from acezero_amd.session import ReconstructionSession
    from tests.test_session_gpu import _opt
    seq = synth.render_room_sequence(seed=2089, n_frames=48, arc_deg=24.0, device="cuda")
    esd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights_bandpass(seed=4099).items()}
    ses = ReconstructionSession(esd, seq["images"], opt=_opt(seq), depth=seq["depth"])
    assert ses.world == 2 and len(ses.owned) == 24
    even = list(range(0, 48, 2))
    m = ses.map(even, seq["poses"][even].cpu(), seq["focal"], iterations=2000, loss_type="tanh", schedule="1cyclepoly", lr_max=0.003)
    poses, inl = ses.register(m["head"], seq["focal"])
    sub_p, sub_i = ses.register(m["head"], seq["focal"], max_estimates=10)
    # fewer images than ranks: ONE rank trains, every rank gets its head (the ranks used to train diverging copies)
    one = ses.map([2], seq["poses"][[2]].cpu(), seq["focal"], iterations=300, loss_type="tanh", schedule="constant", lr_max=0.003, with_depth=True)
    q.put((rank, m["data_parallel"], m["buffer"], m["iterations"], poses, inl, sub_p, sub_i, seq["poses"].cpu().numpy(),
           {k: v.numpy() for k, v in m["head"].items()}, {k: v.numpy() for k, v in one["head"].items()}, one["data_parallel"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_session_maps_and_registers_like_one():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_session_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(_collect(procs, q, 400), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res
    assert a[1] and b[1] and a[2] == b[2] == 24 * 10 * 1024 and a[3] == b[3]
    assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5]), "both ranks hold every frame's pose after the gather"
    assert all(np.array_equal(a[9][k], b[9][k]) for k in a[9]), "replicated heads are bit-identical"
    assert len(a[6]) == 10 and np.array_equal(a[6], b[6])                # --max_estimates: the same random subset on both ranks
    assert not a[11] and not b[11] and all(np.array_equal(a[10][k], b[10][k]) for k in a[10]), "one image on two ranks: one head, handed over"
    poses, inl, gt = a[4], a[5], a[8]
    dt = np.linalg.norm(poses[:, :3, 3] - gt[:, :3, 3], axis=1)
    assert (inl > 500).mean() >= 0.95 and np.median(dt) < 0.02, ((inl > 500).mean(), np.median(dt))


def _sharded_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from acezero_amd import parallel
    prob, flat0 = _problem()
    n = prob["features"].shape[0]
    lo, hi = parallel.shard_range(n, rank, world)
    gen = torch.Generator(device="cuda").manual_seed(8191)
    perm = torch.randperm(n, generator=gen, device="cuda")
    local, offs = parallel.epoch_local_batches(perm, B, lo, hi)
    out = {}
    # the RCCL branch of ShardedDataParallel (one reduce_scatter_tensor, one all_gather_into_tensor on staging buffers) under gloo:
    # gloo has no reduce-scatter, so that one collective is emulated; the slicing, the staging and the kernels around it are real
    def _rs(output, input, op=dist.ReduceOp.SUM, group=None, async_op=False):
        full = input.clone()
        dist.all_reduce(full, op=op, group=group)
        k = output.numel()
        output.copy_(full[dist.get_rank(group) * k:(dist.get_rank(group) + 1) * k])

        class _Done:
            def wait(self):
                return True
        return _Done()
    dist.reduce_scatter_tensor = _rs
    try:
        probe_out, probe_in = torch.zeros(2 * world, dtype=torch.int32, device="cuda"), torch.full((2,), rank, dtype=torch.int32, device="cuda")
        dist.all_gather_into_tensor(probe_out, probe_in)
        assert probe_out.cpu().tolist() == [r for r in range(world) for _ in range(2)]
    except (RuntimeError, NotImplementedError):
        def _ag(output, input, group=None, async_op=False):
            parts = [torch.empty_like(input) for _ in range(dist.get_world_size(group))]
            dist.all_gather(parts, input, group=group)
            output.copy_(torch.cat(parts))
        dist.all_gather_into_tensor = _ag
    for mode in ("allreduce", "sharded", "sharded_oneshot"):
        tr = _make_trainer(prob, flat0, lo, hi)
        dp = parallel.make_data_parallel(tr, mode=mode)
        for b in range(4):
            dp.step(local[offs[b]:offs[b + 1]])
        torch.cuda.synchronize()
        stale = tr.params.clone()
        dp.gather_masters()
        f = torch.from_numpy(prob["features"][:777]).cuda()
        out[mode] = (tr.params.cpu().numpy(), tr.adam_m.cpu().numpy(), tr.adam_v.cpu().numpy(), tr.state(), tr.get_scene_coordinates(f).cpu().numpy(),
                     bool((stale != tr.params).any()))
        tr.close()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_update_equals_the_all_reduce_path_bitwise():
    """parallel.ShardedDataParallel through the real kernels (acez_train_update_layers, export / import of the 16-bit copies with the
    local transpose): two ranks, four steps. A two-term sum has one order, so the reduced gradients -- and with them every
    parameter, both AdamW moments, the schedule state and the scene coordinates computed from the received compute copies --
    must be BIT-identical to the all-reduce + replicated-update path, on both ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(_collect(procs, q, 300))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = res[0]["allreduce"]
    for r in (0, 1):
        for mode in ("allreduce", "sharded", "sharded_oneshot"):
            got = res[r][mode]
            for k in range(3):
                assert np.array_equal(got[k], ref[k]), (r, mode, k)
            assert got[3] == ref[3] and np.array_equal(got[4], ref[4]), (r, mode)
        assert res[r]["sharded"][5] and res[r]["sharded_oneshot"][5] and not res[r]["allreduce"][5]      # masters of the other rank's layers were stale until gathered
    assert ref[3]["iteration"] == 4


def _rccl_world1_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    probe = torch.ones(4, device="cuda")
    dist.all_reduce(probe)
    torch.cuda.synchronize()
    q.put(("group_formed", float(probe.sum())))      # the parent skips the test if this never arrives (no RCCL bootstrap on the box)
    from acezero_amd import parallel
    prob, flat0 = _problem()
    n = prob["features"].shape[0]
    gen = torch.Generator(device="cuda").manual_seed(8191)
    perm = torch.randperm(n, generator=gen, device="cuda")
    local, offs = parallel.epoch_local_batches(perm, B, 0, n)
    out = {}
    for name, kw in (("short_cut", {}), ("rccl", {"force_exchange": True}), ("rccl_per_owner", {"force_exchange": True, "one_shot": False})):
        tr = _make_trainer(prob, flat0, 0, n)
        dp = parallel.ShardedDataParallel(tr, **kw)
        if kw:
            assert dp.exchange and dp.one_shot == kw.get("one_shot", True), (dp.exchange, dp.one_shot)
        for b in range(4):
            dp.step(local[offs[b]:offs[b + 1]])
        torch.cuda.synchronize()
        dp.gather_masters()
        f = torch.from_numpy(prob["features"][:777]).cuda()
        out[name] = (tr.params.cpu().numpy(), tr.adam_m.cpu().numpy(), tr.adam_v.cpu().numpy(), tr.state(), tr.get_scene_coordinates(f).cpu().numpy())
        tr.close()
    # the all-reduce mode's one collective, and the registration gather, on the same backend
    tr = _make_trainer(prob, flat0, 0, n)
    dpa = parallel.DataParallelTrainer(tr, force_exchange=True)   # the default mode's one collective: with one rank the sum is the rank's own gradient
    assert dpa.exchange
    for b in range(4):
        dpa.step(local[offs[b]:offs[b + 1]])
    torch.cuda.synchronize()
    out["allreduce"] = (tr.params.cpu().numpy(), tr.adam_m.cpu().numpy(), tr.adam_v.cpu().numpy(), tr.state())
    tr.close()
    q.put((0, out, dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_branch_of_the_exchange_runs_on_this_stack_with_one_rank():
    """The ONLY hardware run of the RCCL calls a one-GPU box allows: a one-rank "nccl" group, ShardedDataParallel(force_exchange=True):
    async all_reduce of the small bucket, reduce_scatter_tensor / all_gather_into_tensor on the staging buffers (and the per-owner
    reduce / broadcast form), the stream hand-over between RCCL's stream and the launch stream of libacez, import16. With one rank
    every collective is the identity, so parameters, moments, schedule state and scene coordinates must equal the short cut bitwise."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(_free_port(), q))
    p.start()
    import queue
    try:
        first = q.get(timeout=150)
    except queue.Empty:
        first = None
    if first is None or first[0] != "group_formed":
        if p.is_alive():
            p.terminate()
        pytest.skip("a one-rank RCCL group could not be formed on this box within 150 s (exit code %s): the RCCL branch stays covered by the "
                    "gloo-emulated test above" % p.exitcode)
    assert first[1] == 4.0
    (_, out, backend), = _collect([p], q, 300)
    p.join(timeout=60)
    assert p.exitcode == 0 and backend == "nccl"
    ref = out["short_cut"]
    for mode in ("rccl", "rccl_per_owner"):
        for k in range(3):
            assert np.array_equal(out[mode][k], ref[k]), (mode, k)
        assert out[mode][3] == ref[3] and np.array_equal(out[mode][4], ref[4]), mode
    assert ref[3]["iteration"] == 4
    # all-reduce mode: replicated update through acez_train_update (another launch flow: equal to rounding of the schedule only -- same kernels'
    # arithmetic, so bitwise here as in test_sharded_update_equals_the_all_reduce_path_bitwise)
    for k in range(3):
        assert np.array_equal(out["allreduce"][k], ref[k]), ("allreduce", k)

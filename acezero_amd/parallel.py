"""Multi-GPU host logic (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on MI355X).

The reference has no multi-GPU code at all; this module adds the only collective of the hot
path and the sharding rules:

  T (training)      each rank owns a shard of the feature buffer; per step every rank runs backward on its rows with
                    the loss normalised by the GLOBAL batch, the flat gradient bucket (head gradients + 3 statistics)
                    is summed with ONE all-reduce, and every rank applies the identical optimiser/schedule update, so
                    replicas stay bit-identical without ever broadcasting weights.
  R (registration)  frame i belongs to rank i % world. The random stream is keyed by the frame id, so the result of
                    a frame does not depend on the partition; no collective on the hot path, results are gathered once.

Everything here is backend-agnostic (the trainer object only needs backward / grad / update), which is what the
world-size-2 gloo tests on CPU exercise (tests/test_parallel_cpu.py).
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank` (buffer / image-list sharding; sizes differ by at most 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def frames_of_rank(n_frames, rank, world):
    """Registration: frame ids owned by `rank` (round robin keeps the per-rank load even for any frame order)."""
    return list(range(rank, n_frames, world))


def split_batch_by_owner(global_batch_indices, shard_lo, shard_hi):
    """Reference-exact batch composition under data parallelism: every rank draws the SAME global permutation
    (same generator seed, ace_trainer.py:466) and keeps the indices of each batch that fall into its buffer shard,
    re-based to local row numbers. Because the loss is a sum / B (ace_trainer.py:612-613), the all-reduced gradient
    equals the single-GPU gradient of that batch."""
    m = (global_batch_indices >= shard_lo) & (global_batch_indices < shard_hi)
    return (global_batch_indices[m] - shard_lo).contiguous()


def epoch_local_batches(perm, batch, shard_lo, shard_hi):
    """split_batch_by_owner for a whole epoch at once: `perm` is the epoch's global permutation (every rank draws the same one),
    consumed in consecutive slices of `batch` rows with the last partial slice dropped (ace_trainer.py:473-474). Returns
    (local, offsets): this rank's rows of batch b are local[offsets[b]:offsets[b + 1]] (re-based to the shard, in batch order).
    One host synchronisation per EPOCH (the per-batch counts) instead of one per step."""
    nb = int(perm.numel()) // batch
    p = perm[:nb * batch]
    m = (p >= shard_lo) & (p < shard_hi)
    local = (p[m] - shard_lo).contiguous()
    counts = m.view(nb, batch).sum(dim=1).cpu().tolist()
    offsets = [0]
    for c in counts:
        offsets.append(offsets[-1] + int(c))
    return local, offsets


def rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class DataParallelTrainer:
    """The default exchange (make_data_parallel): backward on the rank's rows, ONE synchronous all-reduce of the flat gradient bucket
    (head gradients + statistics [+ pose-network gradient]), the identical AdamW / schedule update on every rank. Wraps a trainer exposing
    backward(indices), grad and update(). force_exchange: a one-rank group still calls the collective (the RCCL call on a one-GPU box:
    tests/test_dp_gpu.py, bench.py's world-1 leg)."""

    def __init__(self, trainer, group=None, one_shot=None, force_exchange=False):
        self.trainer = trainer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.exchange = self.world > 1 or bool(force_exchange)
        import inspect
        self._announce = "next_indices" in inspect.signature(trainer.update).parameters   # (the CPU tests' stand-in trainer has none)

    def step(self, local_indices, next_local_indices=None):
        """next_local_indices: this rank's rows of the NEXT batch (the very tensor the next call will pass), if known: gathered inside this
        step's optimiser launch (HeadTrainer.update(next_indices))."""
        if local_indices.numel() > 0:
            self.trainer.backward(local_indices)
        else:
            self.trainer.grad.zero_()     # this shard holds no row of the batch: it contributes nothing to the sum
        if self.exchange:
            dist.all_reduce(self.trainer.grad, op=dist.ReduceOp.SUM, group=self.group)
        if next_local_indices is None or not self._announce:
            self.trainer.update()
        else:
            self.trainer.update(next_local_indices)

    def gather_masters(self):
        pass   # every rank updates every parameter: nothing to gather


class ShardedDataParallel:
    """The training step's exchange for strong scaling ("prefer one-shot reduce-scatter / all-gather ...,
    optimizer sharded ZeRO-1 style"), replacing the all-reduce of the whole 8.4 MB bucket + a replicated AdamW over 2.1 M parameters:

      small bucket   biases of the wide layers, fc3, the 4 statistics, the pose-network gradient (~0.3 MB): ALL-REDUCE, update replicated
      weight matrices (8.4 MB fp32): REDUCE-SCATTER by layer -- rank r receives the summed gradient of ITS layers only --
                     rank r runs AdamW on those layers (1 / G of the optimiser's 75 MB of HBM traffic),
      16-bit copies  the updated W of every rank's layers (4.2 MB in all): ALL-GATHER; each rank transposes the received layers
                     locally for the input-gradient GEMMs (no second gather for W^T).

    Bytes on the wire per rank and step: (G-1)/G * (8.4 + 4.2) MB + 0.3 MB instead of 2 (G-1)/G * 8.7 MB. Every rank ends a step with
    identical compute copies, biases, fc3, schedule and pose parameters (the all-reduce result is the same on all ranks), i.e. the
    replicas compute identically; only the fp32 masters / AdamW moments of the weight matrices live on their owner until
    gather_masters() (before a checkpoint is written).

    With RCCL and a layer count divisible by the world size the two collectives are one reduce_scatter_tensor and one
    all_gather_into_tensor (in place on slices of the bucket / the gather buffer); otherwise -- gloo in the CPU tests, 8 layers on 3
    ranks -- one reduce / broadcast per owner. The trainer object needs: L, LAYER_STRIDE, grad, backward(rows), update_layers(lo, hi),
    new_weights16_buffer(), export_weights16 / import_weights16(lo, hi, tensor), master_tensors()."""

    def __init__(self, trainer, group=None, one_shot=None, proxy_world=None, force_exchange=False, async_collectives=False):
        """async_collectives: issue the first phase's collectives with async_op=True (round 5's form). Measured with a one-rank RCCL group
        (tools/dp_host_probe.py): 246 us per step against 215 us with synchronous calls -- an async collective is a round trip through
        torch's RCCL stream, two cross-stream events each; the synchronous ones run in the launch stream's order. Default off.
        force_exchange (tests/test_dp_gpu.py, bench.py's rccl_world1 leg): a ONE-rank group still goes through the whole exchange -- the
        staging copies and every collective call, which with RCCL and one rank are real RCCL launches on this stack -- instead of the
        world == 1 short cut. The only way to run the RCCL branch on a one-GPU box; results are bit-identical to the short cut.
        proxy_world = G (measurement only, bench.py's dp_rank_compute legs): this process plays rank 0 of a G-rank job WITHOUT a process
        group -- the same launches, staging copies and layer ownership as a real rank, every collective skipped -- so that the compute
        half of DESIGN.md section 7's budget is a measurement on one GPU. The parameters it produces are meaningless (partial sums)."""
        self.trainer = trainer
        self.group = group
        self.rank, self.world = rank_world(group)
        self.proxy = proxy_world is not None
        self.async_collectives = bool(async_collectives)
        self.exchange = self.world > 1 or bool(force_exchange)
        if self.proxy:
            assert self.world == 1, "proxy_world is a single-process measurement"
            self.rank, self.world = 0, int(proxy_world)
            self.exchange = True
        self.L, self.stride = int(trainer.L), int(trainer.LAYER_STRIDE)
        self.ranges = [shard_range(self.L, r, self.world) for r in range(self.world)]
        self.lo, self.hi = self.ranges[self.rank]
        backend = "nccl" if self.proxy else (dist.get_backend(group) if self.exchange else "")
        # (one_shot=True under gloo: the CPU test of this branch, with reduce_scatter_tensor emulated -- gloo has none)
        self.one_shot = (backend == "nccl" and self.L % self.world == 0) if one_shot is None else bool(one_shot)
        assert not self.one_shot or self.L % self.world == 0
        self.wbuf = trainer.new_weights16_buffer() if self.exchange else None
        # separate send / receive staging for the one-shot collectives (1 MB + 0.5 MB per owned layer; no aliasing of a collective's
        # input and output)
        self.rs_out = trainer.grad.new_empty((self.hi - self.lo) * self.stride) if self.one_shot else None
        self.w_own = self.wbuf.new_empty((self.hi - self.lo, self.wbuf.shape[1])) if self.one_shot else None
        if self.proxy:   # what the skipped collectives would have delivered: finite stand-ins (a NaN would switch the schedule off and
            if self.rs_out is not None:   # turn the timed steps into no-ops): zero weight gradients, the peers' layers unchanged
                self.rs_out.zero_()
            trainer.export_weights16(0, self.L, self.wbuf)

    def _global(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _small(self, grad):
        wide = grad[:self.L * self.stride].view(self.L, self.stride)
        return wide[:, self.stride - 512:], grad[self.L * self.stride:]

    def step(self, local_indices, next_local_indices=None):
        t = self.trainer     # (next_local_indices: accepted for DataParallelTrainer's signature; this flow gathers at the start of a step)
        if local_indices.numel() > 0:
            t.backward(local_indices)
        else:
            t.grad.zero_()     # this shard holds no row of the batch: it contributes nothing to the sums
        if not self.exchange:
            t.update_layers(0, self.L)
            return
        bias, tail = self._small(t.grad)
        small = torch.cat([bias.reshape(-1), tail])
        wide = t.grad[:self.L * self.stride]
        ao = self.async_collectives
        work = [] if self.proxy else [dist.all_reduce(small, op=dist.ReduceOp.SUM, group=self.group, async_op=ao)]
        if self.proxy:
            pass
        elif self.one_shot:
            work.append(dist.reduce_scatter_tensor(self.rs_out, wide, op=dist.ReduceOp.SUM, group=self.group, async_op=ao))
        else:
            for r, (lo, hi) in enumerate(self.ranges):
                if hi > lo:
                    work.append(dist.reduce(wide[lo * self.stride:hi * self.stride], dst=self._global(r),
                                            op=dist.ReduceOp.SUM, group=self.group, async_op=ao))
        for w in work:
            if w is not None:
                w.wait()
        if self.one_shot:
            wide[self.lo * self.stride:self.hi * self.stride].copy_(self.rs_out)
        # biases / fc3 / statistics / pose gradient: everyone uses the all-reduced values (also the owner, whose reduce-scatter result
        # may differ from them in the last bit: another summation order)
        bias.copy_(small[:self.L * 512].view(self.L, 512))
        tail.copy_(small[self.L * 512:])
        t.update_layers(self.lo, self.hi)
        if self.one_shot:
            t.export_weights16(self.lo, self.hi, self.w_own)
            if not self.proxy:
                dist.all_gather_into_tensor(self.wbuf.view(-1), self.w_own.view(-1), group=self.group)
        elif self.proxy:
            t.export_weights16(self.lo, self.hi, self.wbuf[self.lo:self.hi])
        else:
            t.export_weights16(self.lo, self.hi, self.wbuf[self.lo:self.hi])
            for r, (lo, hi) in enumerate(self.ranges):
                if hi > lo:
                    dist.broadcast(self.wbuf[lo:hi], src=self._global(r), group=self.group)
        if hasattr(t, "import_weights16_all"):      # one launch for all peers' layers (a copy + a transpose launch per peer before)
            t.import_weights16_all(self.lo, self.hi, self.wbuf)
        else:
            for r, (lo, hi) in enumerate(self.ranges):
                if r != self.rank:
                    t.import_weights16(lo, hi, self.wbuf[lo:hi])

    def gather_masters(self):
        """fp32 masters and AdamW moments of every layer on every rank (before state_dict() / a checkpoint)."""
        if self.world == 1:
            return
        for r, (lo, hi) in enumerate(self.ranges):
            if hi > lo:
                for x in self.trainer.master_tensors():
                    dist.broadcast(x[lo * self.stride:hi * self.stride], src=self._global(r), group=self.group)
        if hasattr(self.trainer, "masters_synced"):
            self.trainer.masters_synced()


DEFAULT_DP_MODE = "allreduce"


def dp_mode(mode=None):
    import os
    return (mode or os.environ.get("ACEZ_DP_MODE", DEFAULT_DP_MODE)).lower()


def make_data_parallel(trainer, group=None, mode=None):
    """ACEZ_DP_MODE = "allreduce" (default) | "sharded" | "sharded_oneshot".

    allreduce  ONE synchronous all-reduce of the flat gradient bucket (8.7 MB fp32), AdamW replicated -- BASELINE.json north_star's
               "RCCL all-reduce over xGMI on the head gradients only".
    sharded    reduce-scatter by layer + AdamW on the owned layers + all-gather of the 16-bit weights (ShardedDataParallel): a third
               fewer bytes on the wire and 1 / G of the optimiser traffic, but three collectives and five staging launches per step.
    Why all-reduce is the default (round 6, tools/dp_host_probe.py on one MI355X with a ONE-rank RCCL group, i.e. every cost but the wire
    time; profiles/r06_dp_host_probe.log): split flow 168 us, + one synchronous all_reduce 175 us; sharded flow with its collectives
    skipped 186 us, called synchronously 215 us, called with async_op=True (round 5's form: every async collective is a round trip
    through torch's RCCL stream) 246 us. The sharded exchange saves ~4 MB of ring traffic per step (20-40 us at 100-200 GB/s) and pays
    ~33 us of launches and two more collective latencies for it: a wash at best at G <= 8, and the all-reduce is the simpler path."""
    mode = dp_mode(mode)
    if mode == "allreduce":
        return DataParallelTrainer(trainer, group)
    return ShardedDataParallel(trainer, group, one_shot=True if mode == "sharded_oneshot" else None)


def gather_registrations(local_frame_ids, local_poses, local_inliers, n_frames, group=None, expect=None):
    """Collect per-frame results on every rank in frame order. local_poses [k,4,4] f32, local_inliers [k] i32. `expect`: the frame
    ids that must have been registered by some rank (default: all n_frames)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        order = torch.argsort(torch.as_tensor(local_frame_ids))
        return local_poses[order], local_inliers[order]
    world = dist.get_world_size(group)
    payload = (list(local_frame_ids), local_poses.cpu(), local_inliers.cpu())
    out = [None] * world
    dist.all_gather_object(out, payload, group=group)
    poses = torch.zeros(n_frames, 4, 4, dtype=torch.float32)
    inl = torch.zeros(n_frames, dtype=torch.int32)
    seen = torch.zeros(n_frames, dtype=torch.bool)
    for ids, p, c in out:
        idx = torch.as_tensor(ids, dtype=torch.long)
        poses[idx] = p
        inl[idx] = c
        seen[idx] = True
    need = seen if expect is None else seen[torch.as_tensor(expect, dtype=torch.long)]
    assert bool(need.all()), "some frames were not registered by any rank"
    return poses, inl

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
    """Wraps a trainer exposing backward(indices), grad (flat tensor incl. statistics) and update()."""

    def __init__(self, trainer, group=None):
        self.trainer = trainer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def step(self, local_indices):
        if local_indices.numel() > 0:
            self.trainer.backward(local_indices)
        else:
            self.trainer.grad.zero_()     # this shard holds no row of the batch: it contributes nothing to the sum
        if self.world > 1:
            dist.all_reduce(self.trainer.grad, op=dist.ReduceOp.SUM, group=self.group)
        self.trainer.update()


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

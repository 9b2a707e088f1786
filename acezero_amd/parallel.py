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


class DataParallelTrainer:
    """Wraps a trainer exposing backward(indices), grad (flat tensor incl. statistics) and update()."""

    def __init__(self, trainer, group=None):
        self.trainer = trainer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def step(self, local_indices):
        self.trainer.backward(local_indices)
        if self.world > 1:
            dist.all_reduce(self.trainer.grad, op=dist.ReduceOp.SUM, group=self.group)
        self.trainer.update()


def gather_registrations(local_frame_ids, local_poses, local_inliers, n_frames, group=None):
    """Collect per-frame results on every rank in frame order. local_poses [k,4,4] f32, local_inliers [k] i32."""
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
    assert bool(seen.all()), "some frames were not registered by any rank"
    return poses, inl

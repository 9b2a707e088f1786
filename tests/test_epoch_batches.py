"""Host logic of the training loop's batch stream (acezero_amd.head.epoch_batches / epoch_permutations): TrainerACE.run_epoch's walk
(ace_trainer.py:454-497) with the next batch announced across epoch boundaries. Runs on the CPU generator (the product uses the device's)."""
import numpy as np
import pytest
import torch

from acezero_amd.head import epoch_batches, epoch_permutations


@pytest.mark.parametrize("n", [10240, 300000])   # a seed round's buffer (grouped draws) and one beyond the grouping threshold (torch.randperm)
def test_every_epoch_is_a_permutation(n):
    perms = epoch_permutations(n, 7, torch.device("cpu"))
    seen = []
    for _ in range(3):
        p = next(perms)
        assert p.dtype == torch.int64 and p.shape == (n,)
        assert torch.equal(torch.sort(p).values, torch.arange(n))
        seen.append(p.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    again = epoch_permutations(n, 7, torch.device("cpu"))
    assert torch.equal(next(again), seen[0])          # seeded: the stream repeats


@pytest.mark.parametrize("n,batch", [(10240, 5120), (12000, 5120), (5120, 5120), (40000, 4096)])
def test_pairs_walk_the_epochs_and_announce_across_the_boundary(n, batch):
    pairs = epoch_batches(n, batch, 11, torch.device("cpu"))
    nb = n // batch
    prev_next = None
    for epoch in range(3):
        rows = []
        for b in range(nb):
            cur, nxt = next(pairs)
            assert cur.shape == (batch,) and nxt.shape == (batch,) and cur.is_contiguous() and nxt.is_contiguous()
            if prev_next is not None:   # the announced tensor IS the next call's batch: same storage, same offset (the C side compares pointers)
                assert cur.data_ptr() == prev_next.data_ptr() and torch.equal(cur, prev_next)
            prev_next = nxt
            rows.append(cur)
        rows = torch.cat(rows)
        assert len(torch.unique(rows)) == nb * batch and int(rows.max()) < n     # distinct rows of one permutation; the tail is dropped


def test_a_buffer_smaller_than_a_batch_is_an_error():
    with pytest.raises(ValueError):
        next(epoch_batches(100, 5120, 1, torch.device("cpu")))

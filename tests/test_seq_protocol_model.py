"""CPU: an executable model of rowseq_kernel's hand-off protocol (acezero_amd/csrc/head_kernels.hip: rowseq_kernel, SeqLink;
head_api.hip: launch_rowseq, seq_usable). It restates, in a few lines each, the grid decode (workgroup -> XCD, row tile, column
tile), the per-row-tile counters (8 increments per workgroup and seam, target (base[mt] + layer) * 32), the host's base accounting
(a launch advances the bases of ITS row tiles only; chains longer than 8 layers are split; a launch on an ended schedule does no
work but keeps the counters in step) and a dispatcher with a limited number of CU slots per XCD, and checks over random call
sequences that
  * no workgroup ever starts a layer before all four column tiles of its row tile have finished the layer before (the property the
    counters exist for), whatever the interleaving;
  * every launch drains when workgroups are dispatched in order and an XCD has at least four free slots -- also when the slots
    are shared with a second tenant, i.e. when the workgroups of a launch are NOT all resident at once;
  * the in-order assumption is really needed: with four slots and an adversarial dispatch order the same launch deadlocks.
The kernel itself is tested on the GPU (tests/test_seq_gpu.py); this file pins the reasoning DESIGN.md section 3 gives for it."""
import random

import pytest

SEQ_MAX_LAYERS = 8


def decode(block, mtiles):
    """rowseq_kernel: blockIdx -> (xcd, mt, nt); None for the padding workgroups of the rounded-up grid."""
    per_xcd = (mtiles + 7) >> 3
    jx = block >> 3
    mt = (block & 7) * per_xcd + (jx >> 2)
    return None if mt >= mtiles else (block & 7, mt, jx & 3)


class Device:
    def __init__(self):
        self.flags = [0] * 64          # one counter per row tile, never reset
        self.base = [0] * 64           # the host's copy: seams completed per row tile

    def launch(self, rows, n_layers, rng, slots_per_xcd=32, active=True, in_order=True, tenant=0.0, adversarial=False):
        """One rowseq launch under a random interleaving. Returns False on deadlock."""
        mtiles = (rows + 79) // 80
        assert mtiles <= 64
        grid = 32 * ((mtiles + 7) // 8)
        base = list(self.base)                                 # kernel argument: a snapshot at launch time
        for mt in range(mtiles):
            self.base[mt] += n_layers - 1                      # launch_rowseq: only the row tiles of this batch
        queues = {x: [] for x in range(8)}
        for b in range(grid):
            d = decode(b, mtiles)
            if d:
                queues[d[0]].append(d)
        if adversarial:                                        # one workgroup of each row tile first
            for x in queues:
                queues[x].sort(key=lambda d: (d[2], d[1]))
        done_layer = {}                                        # (mt, nt) -> layers finished
        resident = {x: [] for x in range(8)}
        pending = sum(len(q) for q in queues.values())
        idle_rounds = 0
        while pending:
            progressed = False
            for x in range(8):
                free = slots_per_xcd - len(resident[x]) - (rng.randint(0, int(tenant * slots_per_xcd)) if tenant else 0)
                while free > 0 and queues[x]:
                    k = 0 if (in_order or adversarial) else rng.randrange(len(queues[x]))
                    resident[x].append([queues[x].pop(k), 0])
                    free -= 1
                    progressed = True
                rng.shuffle(resident[x])
                for wg in list(resident[x]):
                    (_, mt, nt), layer = wg
                    if not active:                             # schedule ended: bump the counter for all seams at once, leave
                        self.flags[mt] += 8 * (n_layers - 1)
                        resident[x].remove(wg); pending -= 1; progressed = True
                        continue
                    if rng.random() < 0.3:
                        continue                               # this workgroup is slow right now
                    if layer > 0 and self.flags[mt] < (base[mt] + layer) * 32:
                        continue                               # polling
                    if layer > 0:                              # THE property: all four producers of this row tile are done with layer - 1
                        assert all(done_layer.get((mt, c), 0) >= layer for c in range(4)), (mt, nt, layer)
                    done_layer[(mt, nt)] = layer + 1
                    wg[1] = layer + 1
                    progressed = True
                    if layer + 1 < n_layers:
                        self.flags[mt] += 8                    # 8 waves, one increment each
                    else:
                        resident[x].remove(wg); pending -= 1
            idle_rounds = 0 if progressed else idle_rounds + 1
            if idle_rounds > 200:
                return False
        return True

    def chain(self, rows, layers, rng, **kw):
        """launch_rowseq: chains of more than SEQ_MAX_LAYERS layers become several launches."""
        for i0 in range(0, layers, SEQ_MAX_LAYERS):
            if not self.launch(rows, min(SEQ_MAX_LAYERS, layers - i0), rng, **kw):
                return False
        return True


@pytest.mark.parametrize("seed", range(6))
def test_random_call_sequences_never_read_early_and_always_drain(seed):
    rng = random.Random(seed)
    dev = Device()
    for _ in range(40):
        rows = rng.choice([1, 80, 333, 637, 1000, 2048, 2533, 4097, 5120])
        layers = rng.choice([8, 7, 11, 10, 5, 4, 2])           # forward / input-gradient chains of heads with 1, 2 and 0 blocks
        assert dev.chain(rows, layers, rng, active=rng.random() > 0.15)
    assert all(f == 32 * b for f, b in zip(dev.flags, dev.base))   # counters and host bases stay in step


@pytest.mark.parametrize("seed", range(4))
def test_shared_gpu_partial_residency_drains_with_in_order_dispatch(seed):
    """A second tenant takes a random share of the slots every round and only a few slots exist: the workgroups of a launch are never
    all resident; siblings are adjacent in dispatch order, so the one partially resident group per XCD always completes."""
    rng = random.Random(100 + seed)
    dev = Device()
    for _ in range(10):
        assert dev.chain(rng.choice([637, 2533, 5120]), rng.choice([8, 7]), rng, slots_per_xcd=rng.choice([4, 5, 8]), tenant=0.0)
        assert dev.chain(rng.choice([637, 2533, 5120]), rng.choice([8, 7]), rng, slots_per_xcd=12, tenant=0.6)


def test_out_of_order_dispatch_can_deadlock():
    """Four slots per XCD filled with one workgroup of each of four different row tiles: nobody's siblings can ever start. This is why the
    host only uses the kernel when the whole grid fits the chip, and why DESIGN.md lists in-order dispatch as an assumption."""
    rng = random.Random(7)
    dev = Device()
    assert not dev.launch(5120, 8, rng, slots_per_xcd=4, adversarial=True)

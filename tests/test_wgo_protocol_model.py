"""CPU: an executable model of wgrad_opt_kernel's exchange protocol (acezero_amd/csrc/head_kernels.hip: wgrad_opt_kernel; head_api.hip:
wgrad_opt_usable, the launch in train_backward_impl). It restates the grid decode (workgroup -> XCD, layer, slab, tile), the two
counters of a tile (one per receiving workgroup, bumped once by each of the partner's two sending waves, never reset; a launch waits
for 2 x the number of launches so far) and a dispatcher with a limited number of CU slots per XCD, and checks that
  * the two slabs of every tile run on ONE XCD under the round-robin placement the probe verifies (the hand-off lives in that XCD's L2);
  * over random interleavings of many launches no workgroup passes its poll before BOTH sending waves of its partner have stored in
    THAT launch -- counts left by earlier launches can never satisfy a later target, and the exchange being unconditional (inactive
    schedule, NaN loss, overflow: only the final stores are guarded) keeps the counters of the two partners in step;
  * every launch drains when all its workgroups are resident (the host's condition: grid <= CUs) and, dispatched in order, already
    with 17 free slots per XCD; with 16 an in-order launch deadlocks (the partner of the first workgroup is 16 positions behind it):
    what the bounded poll + fault word + fall-back are for when a second tenant holds CUs.
The kernel itself is tested on the GPU (tests/test_wgrad_opt_gpu.py); this file pins the reasoning of DESIGN.md section 3."""
import random

import pytest


def decode(block, n_layers):
    """wgrad_opt_kernel: blockIdx -> (xcd, layer, slab, tile); None for the padding workgroups of the rounded-up grid."""
    xcd, jx = block & 7, block >> 3
    layer, slab, tile = xcd + 8 * (jx >> 5), (jx >> 4) & 1, jx & 15
    return None if layer >= n_layers else (xcd, layer, slab, tile)


class Device:
    def __init__(self, n_layers):
        self.L = n_layers
        self.flags = {}        # (layer, tile, receiving slab) -> count, never reset
        self.epoch = 0         # host: launches so far

    def launch(self, rng, slots_per_xcd=32, in_order=True):
        """One launch under a random interleaving; returns False on deadlock. Asserts the ordering property on the way."""
        self.epoch += 1
        target = 2 * self.epoch
        grid = 256 * ((self.L + 7) // 8)
        queues = {x: [] for x in range(8)}
        for b in range(grid):
            d = decode(b, self.L)
            if d:
                queues[d[0]].append(d)
        sent = {}              # (layer, tile, sending slab) -> sending waves that have signalled in THIS launch
        resident = {x: [] for x in range(8)}   # [workgroup, state]: 0 = K loop, 1 = sending (two waves, one after the other), 2 = polling, 3 = done
        pending = sum(len(q) for q in queues.values())
        idle = 0
        while pending:
            progressed = False
            for x in range(8):
                while len(resident[x]) < slots_per_xcd and queues[x]:
                    k = 0 if in_order else rng.randrange(len(queues[x]))
                    resident[x].append([queues[x].pop(k), 0, 0])
                    progressed = True
                rng.shuffle(resident[x])
                for wg in resident[x]:
                    (_, layer, slab, tile), state, nsent = wg
                    if rng.random() < 0.5:
                        continue
                    if state == 0:
                        wg[1] = 1
                        progressed = True
                    elif state == 1:                      # one sending wave: stores acknowledged, then the partner's counter
                        key = (layer, tile, 1 - slab)
                        self.flags[key] = self.flags.get(key, 0) + 1
                        sent[(layer, tile, slab)] = sent.get((layer, tile, slab), 0) + 1
                        wg[2] = nsent + 1
                        if wg[2] == 2:
                            wg[1] = 2
                        progressed = True
                    elif state == 2:                      # the loader waves' poll
                        if self.flags.get((layer, tile, slab), 0) - target >= 0:
                            assert sent.get((layer, tile, 1 - slab), 0) == 2, "poll passed before the partner had sent in this launch"
                            wg[1] = 3
                            progressed = True
                done = [wg for wg in resident[x] if wg[1] == 3]
                pending -= len(done)
                resident[x] = [wg for wg in resident[x] if wg[1] != 3]
            idle = 0 if progressed else idle + 1
            if idle > 200:
                return False
        return True


def test_both_slabs_of_a_tile_share_an_xcd():
    for L in range(1, 9):
        seen = {}
        for b in range(256 * ((L + 7) // 8)):
            d = decode(b, L)
            if d:
                seen.setdefault((d[1], d[3]), set()).add(d[0])
                assert d[0] == b % 8
        assert len(seen) == L * 16 and all(len(v) == 1 for v in seen.values())


@pytest.mark.parametrize("seed", range(6))
def test_no_poll_passes_early_and_resident_launches_drain(seed):
    rng = random.Random(seed)
    dev = Device(8)
    for _ in range(12):
        assert dev.launch(rng, slots_per_xcd=32, in_order=rng.random() < 0.5)
    # counters are monotone and in step with the host's epoch: exactly two increments per receiving workgroup and launch
    assert all(v == 2 * dev.epoch for v in dev.flags.values()) and len(dev.flags) == 8 * 16 * 2


def test_in_order_dispatch_needs_seventeen_slots():
    rng = random.Random(3)
    assert Device(8).launch(rng, slots_per_xcd=17, in_order=True)
    assert not Device(8).launch(rng, slots_per_xcd=16, in_order=True)

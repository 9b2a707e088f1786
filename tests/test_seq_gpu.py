"""GPU: rowseq_kernel (head_kernels.hip; the forward layers / the input-gradient layers in ONE launch each, the kernel
boundaries replaced by a same-XCD hand-off; the default) against the per-layer rowgemm80 launches (ACEZ_SEQ=0) on identical
inputs. Same tiles, same ring, same K order, same epilogues, same row groups for the bias partials, so EVERYTHING must agree bit for
bit: activations, propagated gradients, the full gradient vector, parameters and optimiser state after free-running steps. The
hand-off is a race if it is wrong, so the comparison is also run over a few hundred steps and over ragged / tiny batches."""
import os

import numpy as np
import pytest
import torch

from tests import helpers
from tests.helpers import big_problem as _big_problem
from tests.test_head_gpu import _trainer

pytestmark = pytest.mark.gpu


def _pair(make):
    out = []
    for seq in ("0", "1"):
        os.environ["ACEZ_SEQ"] = seq
        try:
            out.append(make())
        finally:
            os.environ.pop("ACEZ_SEQ", None)
    return out


@pytest.mark.parametrize("name,n", [("head_tanh_1cyclepoly", 5120), ("head_tanh_1cyclepoly", 80), ("head_dyntanh_circle", 637),
                                    ("head_tanh_calib", 1000), ("head_tanh_posemlp", 2048), ("head_tanh_depth", 4097)])
def test_one_launch_chains_equal_per_layer_launches(name, n):
    prob = _big_problem()
    from oracle import head_oracle
    flat0 = head_oracle.init_params(helpers.SEED + 1)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS[name], prob)
    cfg["global_batch"] = n
    ref, new = _pair(lambda: _trainer(prob, flat0, cfg, max_batch=5120))
    rng = np.random.default_rng(5)
    L = ref.L
    for it in range(4):
        idx = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda()
        ref.backward(idx)
        new.backward(idx)
        torch.cuda.synchronize()
        for l in range(L):
            if not (l < L - 2 and l % 3 == 2):   # (a block's last activation is not kept in training: only its ReLU mask bits are)
                assert np.array_equal(ref.debug_read("out", l, n), new.debug_read("out", l, n)), ("out", l, it)
            if l != L - 1:                       # (fc2's mask is applied by the loss kernel from the activation itself)
                assert np.array_equal(ref.debug_read("mask", l, n), new.debug_read("mask", l, n)), ("mask", l, it)
            assert np.array_equal(ref.debug_read("dZ", l, n), new.debug_read("dZ", l, n)), ("dZ", l, it)
        for b in range(ref.nb + 2):
            assert np.array_equal(ref.debug_read("R", b, n), new.debug_read("R", b, n)), ("R", b, it)
        assert torch.equal(ref.grad, new.grad), it
        ref.update()
        new.update()
        torch.cuda.synchronize()
        assert torch.equal(ref.params, new.params) and ref.state() == new.state()


def _decode_mask_bits(words, n):
    """[row tiles, 2048] uint32 words of a layer (RowGemmArgs::mask_out) -> bool [n, 512]: lane-private layout of rowgemm80's accumulators."""
    mt_n = words.shape[0]
    w4 = words.reshape(mt_n, 4, 4, 64, 2)                     # [row tile][column tile][wave][lane][x | y]
    out = np.zeros((mt_n * 80, 512), bool)
    lane = np.arange(64)
    fr, fq = lane & 15, lane >> 4
    for j in range(5):
        for i in range(2):
            f = 2 * j + i
            for nt in range(4):
                for w in range(4):
                    rows = (np.arange(mt_n)[:, None] * 80 + j * 16 + fr[None, :])              # [mt, lane]
                    col0 = nt * 128 + w * 32 + i * 16 + 4 * fq                                 # [lane]
                    for word, cbase in ((0, 0), (1, 2)):
                        v = w4[:, nt, w, :, word]
                        out[rows, (col0 + cbase)[None, :]] = (v >> (9 - f)) & 1
                        out[rows, (col0 + cbase + 1)[None, :]] = (v >> (25 - f)) & 1
    return out[:n]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("n", [5120, 637])
def test_relu_mask_bits_are_the_sign_of_the_stored_activations(n, dtype):
    """Round 6: the input-gradient layers take their ReLU masks as 40 lane-private bits per multiplier lane instead of re-reading the
    forward layer's 16-bit tile. The bits a training forward leaves, decoded through the accumulator layout, must be exactly `activation
    > 0` of the tile the same forward stored (every layer whose activation is kept), in both operand formats and with a ragged last tile."""
    prob = _big_problem()
    from oracle import head_oracle
    flat0 = head_oracle.init_params(helpers.SEED + 1)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    cfg["global_batch"] = n
    tr = _trainer(prob, flat0, cfg, max_batch=5120, dtype=dtype)
    idx = torch.from_numpy(np.random.default_rng(3).permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda()
    tr.backward(idx)
    torch.cuda.synchronize()
    L = tr.L
    for l in range(L - 1):
        if l < L - 2 and l % 3 == 2:
            continue                                   # pre-residual activation: not kept in training
        act = tr.debug_read("out", l, n)               # 16-bit patterns
        positive = (act != 0) & (act < 0x8000)
        bits = _decode_mask_bits(tr.debug_read("mask", l, n), n)
        assert np.array_equal(bits, positive), (l, int((bits != positive).sum()))
        assert 0.05 < positive.mean() < 0.95           # (a real mask, not all-ones / all-zeros)


def test_free_running_steps_stay_bitwise_equal():
    """300 fused steps at BASELINE's batch: a hand-off that let a workgroup read a tile early would show up as a diverging trajectory."""
    prob = _big_problem()
    from oracle import head_oracle
    flat0 = head_oracle.init_params(helpers.SEED + 1)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    cfg["global_batch"] = 5120
    cfg["iterations"] = 400
    ref, new = _pair(lambda: _trainer(prob, flat0, cfg, max_batch=5120))
    rng = np.random.default_rng(11)
    batches = [torch.from_numpy(rng.permutation(prob["features"].shape[0])[:5120].astype(np.int64)).cuda() for _ in range(300)]
    for tr in (ref, new):
        for idx in batches:
            tr.step(idx)
    torch.cuda.synchronize()
    assert torch.equal(ref.params, new.params) and torch.equal(ref.adam_m, new.adam_m) and torch.equal(ref.adam_v, new.adam_v)
    assert ref.state() == new.state()


def test_more_layers_than_one_launch_holds_and_inference():
    """num_head_blocks = 2: 11 forward / 10 input-gradient layers = two launches of the sequence kernel each; 0 blocks; the plain
    3-channel head; and the inference entry (frame-sized batches take the sequence kernel, larger ones the per-layer path)."""
    from acezero_amd.head import HeadTrainer
    from acezero_amd import synth
    prob = _big_problem(n_images=8, patches_per_view=256)
    n = 1024
    for nb, homog in ((2, True), (0, True), (1, False)):
        def make():
            tr = HeadTrainer(prob["mean"], num_head_blocks=nb, use_homogeneous=homog, max_batch=n, loss_type="tanh", schedule="constant",
                             iterations=50, lr_min=3e-4)
            tr.load_flat(torch.from_numpy(synth.init_head_params(11, num_head_blocks=nb, use_homogeneous=homog)))
            tr.set_buffer(prob["features"], prob["target_px"], prob["view_idx"], prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"],
                          prob["view_image"], prob["image_pose_inv"])
            return tr
        ref, new = _pair(make)
        rng = np.random.default_rng(3)
        for it in range(5):
            idx = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda()
            ref.step(idx)
            new.step(idx)
        torch.cuda.synchronize()
        assert torch.equal(ref.params, new.params), (nb, homog)
        for rows in (1, 333, 1024, 3000):
            f = torch.from_numpy(prob["features"][:rows]).cuda()
            assert torch.equal(ref.get_scene_coordinates(f), new.get_scene_coordinates(f)), (nb, homog, rows)


def test_counters_keep_step_when_training_has_ended_and_batch_sizes_change():
    """The hand-off counters are per row tile and the host keeps their bases: steps past the end of the schedule (no-ops on the
    device), then smaller and larger inference batches, must neither hang nor differ."""
    from acezero_amd.head import HeadTrainer
    from acezero_amd import synth
    prob = _big_problem(n_images=8, patches_per_view=256)

    def make():
        tr = HeadTrainer(prob["mean"], num_head_blocks=1, use_homogeneous=True, max_batch=2048, loss_type="tanh", schedule="constant",
                         iterations=3, lr_min=3e-4)
        tr.load_flat(torch.from_numpy(synth.init_head_params(11, num_head_blocks=1, use_homogeneous=True)))
        tr.set_buffer(prob["features"], prob["target_px"], prob["view_idx"], prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"],
                      prob["view_image"], prob["image_pose_inv"])
        return tr
    ref, new = _pair(make)
    rng = np.random.default_rng(9)
    for n in (2048, 2048, 700, 2048, 100, 1500):   # the schedule ends after 3 of them
        idx = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda()
        ref.step(idx)
        new.step(idx)
        f = torch.from_numpy(prob["features"][:n // 3 + 1]).cuda()
        assert torch.equal(ref.get_scene_coordinates(f), new.get_scene_coordinates(f)), n
    torch.cuda.synchronize()
    assert torch.equal(ref.params, new.params) and ref.state()["iteration"] == 3


def test_sibling_workgroups_share_an_xcd(diag_lib):
    """The hand-off is only valid inside one XCD's L2: the four column tiles of every row tile must have run on the same XCD
    (HW_REG_XCC_ID recorded by the kernel, ACEZ_SEQ_XCC=1; tools/seq_stress.py checks the same with two processes on the GPU)."""
    import ctypes as C
    from acezero_amd import _native as N
    prob = _big_problem(n_images=8, patches_per_view=256)
    from oracle import head_oracle
    flat0 = head_oracle.init_params(helpers.SEED + 1)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    cfg["global_batch"] = 4096
    os.environ["ACEZ_SEQ_XCC"] = "1"
    try:
        tr = _trainer(prob, flat0, cfg, max_batch=4096)
    finally:
        os.environ.pop("ACEZ_SEQ_XCC", None)
    rng = np.random.default_rng(2)
    for n in (4096, 333, 2000):
        tr.backward(torch.from_numpy(rng.permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda())
        torch.cuda.synchronize()
        rec = np.zeros(8 + 256, np.uint32)
        N.check(tr.lib.acez_trainer_debug_read(tr._h, 6, 0, rec.ctypes.data_as(C.c_void_p), rec.nbytes, None))
        sib = rec[8:8 + 4 * ((n + 79) // 80)].reshape(-1, 4)
        assert (sib == sib[:, :1]).all(), (n, sib[(sib != sib[:, :1]).any(1)][:4])
        assert (sib < 8).all()


def _small_trainer(prob, iterations=50, max_batch=2048):
    from acezero_amd.head import HeadTrainer
    from acezero_amd import synth
    tr = HeadTrainer(prob["mean"], num_head_blocks=1, use_homogeneous=True, max_batch=max_batch, loss_type="tanh", schedule="constant",
                     iterations=iterations, lr_min=3e-4)
    tr.load_flat(torch.from_numpy(synth.init_head_params(11, num_head_blocks=1, use_homogeneous=True)))
    tr.set_buffer(prob["features"], prob["target_px"], prob["view_idx"], prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"],
                  prob["view_image"], prob["image_pose_inv"])
    return tr


def test_placement_probe_passes_on_this_gpu_and_gates_the_chains():
    """acez_trainer_create runs a launch with rowseq_kernel's geometry and enables the one-launch chains only if the four column
    tiles of every row tile ran on one XCD (HIP does not promise the mapping). On an MI355X in SPX mode the probe passes."""
    prob = _big_problem(n_images=8, patches_per_view=256)
    tr = _small_trainer(prob)
    st = tr.seq_status()
    assert st == {"enabled": True, "probe": 1, "faults": 0}, st
    os.environ["ACEZ_SEQ"] = "0"
    try:
        assert _small_trainer(prob).seq_status()["enabled"] is False
    finally:
        os.environ.pop("ACEZ_SEQ", None)


@pytest.mark.parametrize("fault_at", [2, 3])   # launch 2 = the forward chain of the second step, 3 = its input-gradient chain
def test_expired_handoff_poll_abandons_the_step_and_falls_back_to_per_layer_launches(fault_at, diag_lib):
    """Fault injection (ACEZ_SEQ_FAULT_AT): one seam of one launch waits for a count that never comes -- what a sibling on a foreign
    XCD looks like. The poll must expire (no hang), the step it happened in must leave parameters, optimiser state and iteration
    count untouched, every later step before the next state read must be a no-op too, state() must switch the trainer to
    per-layer launches, and from there on the trajectory must be bit-identical to a per-layer trainer fed the surviving batches."""
    prob = _big_problem(n_images=8, patches_per_view=256)
    os.environ.update(ACEZ_SEQ_FAULT_AT=str(fault_at), ACEZ_SEQ_SPIN_US="3000")
    try:
        new = _small_trainer(prob)
    finally:
        os.environ.pop("ACEZ_SEQ_FAULT_AT"); os.environ.pop("ACEZ_SEQ_SPIN_US")
    os.environ["ACEZ_SEQ"] = "0"
    try:
        ref = _small_trainer(prob)
    finally:
        os.environ.pop("ACEZ_SEQ", None)
    rng = np.random.default_rng(4)
    batches = [torch.from_numpy(rng.permutation(prob["features"].shape[0])[:2048].astype(np.int64)).cuda() for _ in range(7)]
    new.step(batches[0])
    ref.step(batches[0])
    torch.cuda.synchronize()
    assert torch.equal(new.params, ref.params)
    before = (new.params.clone(), new.adam_m.clone(), new.adam_v.clone())
    new.step(batches[1])          # the faulting step
    new.step(batches[2])          # issued before the host knows: must be a no-op as well
    torch.cuda.synchronize()
    assert torch.equal(new.params, before[0]) and torch.equal(new.adam_m, before[1]) and torch.equal(new.adam_v, before[2])
    st = new.state()              # the state read performs the fall-back
    assert st["iteration"] == 1 and not st["nan"], st
    assert new.seq_status() == {"enabled": False, "probe": 1, "faults": 1}
    for b in batches[3:]:
        new.step(b)
        ref.step(b)
    torch.cuda.synchronize()
    assert torch.equal(new.params, ref.params) and torch.equal(new.adam_m, ref.adam_m) and torch.equal(new.adam_v, ref.adam_v)
    assert new.state() == ref.state()
    f = torch.from_numpy(prob["features"][:777]).cuda()
    assert torch.equal(new.get_scene_coordinates(f), ref.get_scene_coordinates(f))


@pytest.mark.parametrize("announce", [False, True])
def test_fault_word_travels_in_the_gradient_bucket(announce, diag_lib):
    """Data-parallel flow (backward / all-reduce / update): statistics slot 3 of the bucket carries the rank's fault word, so that
    after the all-reduce EVERY rank skips the optimiser step and the replicas stay identical. One process plays both ranks here:
    rank A faults, rank B does not; B receives A's slot through the (emulated) sum and must skip its update and fall back too."""
    prob = _big_problem(n_images=8, patches_per_view=256)
    os.environ.update(ACEZ_SEQ_FAULT_AT="0", ACEZ_SEQ_SPIN_US="3000")
    try:
        a = _small_trainer(prob)
    finally:
        os.environ.pop("ACEZ_SEQ_FAULT_AT"); os.environ.pop("ACEZ_SEQ_SPIN_US")
    b = _small_trainer(prob)
    rng = np.random.default_rng(8)
    idx = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:2048].astype(np.int64)).cuda()
    a.backward(idx)
    b.backward(idx)
    torch.cuda.synchronize()
    assert float(a.grad[a.n_params + 3]) == 1.0 and float(b.grad[b.n_params + 3]) == 0.0
    total = a.grad + b.grad       # the all-reduce
    a.grad.copy_(total)
    b.grad.copy_(total)
    p0 = b.params.clone()
    # announce: acez_train_update_next -- the schedule wave that closes the step sits in the SAME launch as the optimiser workgroup that
    # raises the local fault word from the all-reduced slot; it must decide on the slot itself (no iteration counted on either rank)
    nxt = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:2048].astype(np.int64)).cuda() if announce else None
    a.update(nxt)
    b.update(nxt)
    torch.cuda.synchronize()
    assert torch.equal(b.params, p0) and torch.equal(a.params, p0)
    if announce:     # a step issued before the host knows (with the announced rows): a no-op on both
        a.backward(nxt); b.backward(nxt)
        torch.cuda.synchronize()
        assert torch.equal(b.params, p0) and torch.equal(a.params, p0)
    assert a.state()["iteration"] == 0 and b.state()["iteration"] == 0
    assert a.seq_status()["faults"] == 1 and b.seq_status()["faults"] == 1 and not b.seq_status()["enabled"]


@pytest.mark.parametrize("mask", ["", "0:0-223", "0:0-7,16-255", "0:0-31,64-255"])
def test_restricted_cu_sets_are_bit_identical_or_fall_back(mask):
    """Foreign placement forced from outside (HSA_CU_MASK leaves some XCDs short of CUs, or the runtime reports fewer CUs): the
    trainer must either not use the chains (probe), or be bit-identical with them, or fall back after a bounded poll --
    tools/seq_cu_mask.py checks the surviving trajectory against per-layer launches. Never a hang, never a silent difference."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("ACEZ_SEQ", None)
    if mask:
        env["HSA_CU_MASK"] = mask
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "seq_cu_mask.py"), "9", "5120"], env=env, capture_output=True, text=True,
                       timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    rec = json.loads(lines[-1])
    print("HSA_CU_MASK=%r -> %s" % (mask, rec))
    assert rec["bit_identical"], rec
    if not mask:
        assert rec["probe"] == 1 and rec["enabled_at_end"] and rec["faults"] == 0, rec

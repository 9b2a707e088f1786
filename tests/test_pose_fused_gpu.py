"""GPU: --pose_refinement mlp folded into the step's own launches (acezero_amd/csrc/pose_fused.hip: the pose network's forward as
the first workgroups of the gather launch, its reduction + backward chain as the first workgroups of the head's AdamW launch, its
weight gradients with AdamW in the tile epilogue) against the separate launches of round 2 (ACEZ_POSE_FUSED=0) and against the
split backward / update flow a data-parallel host uses. The per-image sums run in the same order and the optimiser arithmetic is
one un-contracted function inlined at every site, so everything must agree BIT FOR BIT: refined poses, pose gradients, pose
parameters and their AdamW state, and the head's parameters. Reference: refine_poses.py:152-176,212-244, ace_trainer.py:620-640."""
import os

import numpy as np
import pytest
import torch

from oracle import head_oracle
from tests import helpers
from tests.test_head_gpu import _trainer

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _diagnostics_build(diag_lib):
    """ACEZ_POSE_FUSED / ACEZ_POSE_TILE (the side-stream launches and the other tile sizes) are switches of the diagnostics build."""
    yield


def _problem(n_images, patches_per_view=128):
    from acezero_amd import synth
    return synth.make_training_problem(seed=11, n_images=n_images, views_per_image=2, patches_per_view=patches_per_view)


def _mk(prob, cfg, fused, max_batch, tile="16"):
    os.environ["ACEZ_POSE_FUSED"] = fused
    os.environ["ACEZ_POSE_TILE"] = tile
    try:
        return _trainer(prob, head_oracle.init_params(helpers.SEED + 1), cfg, max_batch=max_batch)
    finally:
        os.environ.pop("ACEZ_POSE_FUSED", None)
        os.environ.pop("ACEZ_POSE_TILE", None)


def _same(a, b):
    return (torch.equal(a.params, b.params) and torch.equal(a.adam_m, b.adam_m) and torch.equal(a.adam_v, b.adam_v) and
            torch.equal(a.pose_params, b.pose_params) and torch.equal(a.pose_m, b.pose_m) and torch.equal(a.pose_v, b.pose_v))


@pytest.mark.parametrize("name,n_images,n", [("head_tanh_posemlp", 37, 1024), ("head_tanh_posemlp", 1000, 5120),
                                             ("head_tanh_posemlp_procrustes", 16, 333), ("head_tanh_posemlp", 3, 2048)])
def test_fused_pose_launches_equal_separate_launches_bitwise(name, n_images, n):
    """37 images: ragged last tile; 1000 images at batch 5120: BASELINE's refinement step; 16: exactly one tile; 3 images and 2048 rows:
    ~680 rows per image overflow the one-pass hit lists (the multi-pass fall-back of the reduction)."""
    prob = _problem(n_images, patches_per_view=max(128, 2 * n // (2 * n_images) + 1))
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS[name], prob)
    cfg.update(global_batch=n, pose_refinement_wait=2, refine_calibration=True)
    old, new, split = _mk(prob, cfg, "0", n), _mk(prob, cfg, "1", n), _mk(prob, cfg, "1", n)
    rng = np.random.default_rng(5)
    N = prob["features"].shape[0]
    for it in range(6):
        idx = torch.from_numpy(rng.permutation(N)[:min(n, N)].astype(np.int64)).cuda()
        for tr in (old, new):
            tr.step(idx)
        split.backward(idx)
        g_split = split.grad.clone()
        split.update()
        torch.cuda.synchronize()
        assert _same(old, new), (it, "fused vs separate launches")
        assert _same(new, split), (it, "fused single-GPU step vs backward / update")
        np.testing.assert_array_equal(old.current_poses(), new.current_poses())
        # the pose gradient the data-parallel host all-reduces is the one the fused step applied
        assert torch.equal(new.grad[new.n_params + 4:], g_split[new.n_params + 4:]), it
        assert old.state() == new.state() == split.state()
    moved = (new.pose_params.cpu() - torch.as_tensor(helpers_pose0(new))).abs().max()
    assert float(moved) > 0        # the pose optimiser did step after the wait


@pytest.mark.parametrize("tile", ["8", "4"])
@pytest.mark.parametrize("name,n_images,n", [("head_tanh_posemlp", 37, 1024), ("head_tanh_posemlp", 1000, 5120),
                                             ("head_tanh_posemlp_procrustes", 21, 333), ("head_tanh_posemlp", 3, 2048)])
def test_small_image_tiles_match_the_16_image_tiles(tile, name, n_images, n):
    """The default tile of the fused path is 4 images (pose_small.hip: v_mfma_f32_4x4x1_16b_f32, the reduction split in four slices in
    the forward and two in the backward chain, combined in a fixed order), a different -- equally valid -- summation order than the
    16-image tiles: refined poses and pose gradients agree to fp32 rounding, not bit for bit (3e-5 of the gradient's norm: seven
    128-deep fp32 reductions each way, and an activation within rounding of zero flips its relu mask). What must still hold bit for bit is that the fused single-GPU step and the
    split backward / update flow of the SAME tile size agree (same kernels' bodies), and two runs (determinism)."""
    prob = _problem(n_images, patches_per_view=max(128, 2 * n // (2 * n_images) + 1))
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS[name], prob)
    cfg.update(global_batch=n, pose_refinement_wait=1, refine_calibration=True)
    ref, new, split, again = _mk(prob, cfg, "1", n, "16"), _mk(prob, cfg, "1", n, tile), _mk(prob, cfg, "1", n, tile), _mk(prob, cfg, "1", n, tile)
    rng = np.random.default_rng(6)
    N = prob["features"].shape[0]
    npar = ref.n_params
    for it in range(4):
        idx = torch.from_numpy(rng.permutation(N)[:min(n, N)].astype(np.int64)).cuda()
        # every step is compared in isolation: the small-tile trainers start it from the 16-image trainer's state
        for tr in (new, split, again):
            for a, b in ((tr.params, ref.params), (tr.adam_m, ref.adam_m), (tr.adam_v, ref.adam_v), (tr.pose_params, ref.pose_params),
                         (tr.pose_m, ref.pose_m), (tr.pose_v, ref.pose_v)):
                a.copy_(b)
            tr.sync_weights()
        np.testing.assert_allclose(new.current_poses(), ref.current_poses(), rtol=0, atol=2e-6)
        ref.backward(idx)
        split.backward(idx)
        torch.cuda.synchronize()
        g_ref, g_new = ref.grad[npar + 4:].cpu().numpy(), split.grad[npar + 4:].cpu().numpy()
        assert np.linalg.norm(g_new - g_ref) <= 3e-5 * np.linalg.norm(g_ref), (it, np.linalg.norm(g_new - g_ref) / np.linalg.norm(g_ref))
        h_ref, h_new = ref.grad[:npar].cpu().numpy(), split.grad[:npar].cpu().numpy()   # (the head sees refined poses that differ in the last bits)
        assert np.linalg.norm(h_new - h_ref) <= 2e-3 * np.linalg.norm(h_ref)
        ref.update()
        split.update()
        new.step(idx)
        again.step(idx)
        torch.cuda.synchronize()
        assert _same(new, split), (it, "fused step vs backward / update, tile " + tile)
        assert _same(new, again), (it, "two runs differ")


def helpers_pose0(tr):
    from acezero_amd.head import init_pose_network
    return init_pose_network(helpers.SEED + 3).numpy()


def test_loaded_pose_parameters_reach_the_fused_forward():
    """The fused forward reads transposed weight copies that the fused optimiser epilogue keeps current; parameters written from
    outside (checkpoint load) followed by sync_weights() must be picked up too."""
    prob = _problem(20)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_posemlp"], prob)
    cfg.update(global_batch=512)
    a, b = _mk(prob, cfg, "1", 512), _mk(prob, cfg, "0", 512)
    idx = torch.from_numpy(np.arange(512, dtype=np.int64)).cuda()
    for tr in (a, b):
        tr.step(idx)
    g = torch.Generator().manual_seed(3)
    newp = (torch.rand(a.pose_params.numel(), generator=g) * 0.2 - 0.1).cuda()
    for tr in (a, b):
        tr.pose_params.copy_(newp)
        tr.sync_weights()
        tr.step(idx)
    torch.cuda.synchronize()
    assert _same(a, b)
    np.testing.assert_array_equal(a.current_poses(), b.current_poses())


def test_refinement_step_with_the_next_batch_announced_equals_plain_steps_bitwise():
    """Round 5: with pose refinement folded into the step's launches the NEXT batch is gathered beside the loss kernel too (into the
    trainer's second input buffer; the step's own launch then carries the pose network's forward and the schedule wave only). Head and
    pose-network parameters, all moments, the schedule state and the log must equal plain acez_train_step calls bit for bit -- also
    when the announcement is wrong, when a state read or a split step comes in between, and with ragged batch sizes."""
    from tests.helpers import big_problem as _big_problem
    from tests.test_head_gpu import _trainer
    from oracle import head_oracle
    prob = _big_problem(n_images=8, patches_per_view=512)
    flat0 = head_oracle.init_params(helpers.SEED + 1)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_posemlp"], prob)
    cfg.update(global_batch=2048, iterations=60, refine_calibration=True)
    plain, piped = (_trainer(prob, flat0, cfg, max_batch=2048) for _ in range(2))
    rng = np.random.default_rng(19)
    N = prob["features"].shape[0]
    batches = [torch.from_numpy(rng.permutation(N)[:(2048 if i % 5 else 1111)].astype(np.int64)).cuda() for i in range(24)]
    other = torch.from_numpy(rng.permutation(N)[:2048].astype(np.int64)).cuda()
    for i, b in enumerate(batches):
        plain.step(b)
        nxt = batches[i + 1] if i + 1 < len(batches) else None
        if i % 7 == 3:
            nxt = other                       # a wrong announcement: the next call must notice and gather its own batch
        if i % 11 == 5:
            piped.backward(b); piped.update()   # a split step in between
        else:
            piped.step(b, nxt)
        if i % 9 == 4:
            assert plain.state() == piped.state()
    torch.cuda.synchronize()
    assert torch.equal(plain.params, piped.params) and torch.equal(plain.adam_m, piped.adam_m) and torch.equal(plain.adam_v, piped.adam_v)
    assert torch.equal(plain.pose_params, piped.pose_params) and torch.equal(plain.pose_m, piped.pose_m) and torch.equal(plain.pose_v, piped.pose_v)
    sp, sq = plain.state(), piped.state()
    assert sp == sq and sp["iteration"] == 24
    lp, lq = plain.log(0, 24), piped.log(0, 24)
    assert np.array_equal(lp[0], lq[0]) and np.array_equal(lp[1], lq[1])


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_split_refinement_step_folded_and_announced_equals_fused_steps_bitwise(dtype):
    """Round 6, the data-parallel rank's refinement step (backward / all-reduce / update with --pose_refinement mlp): S1 rides at the front
    of the gradient-reduction launch, the pose parameters' AdamW and the refresh of the transposed copies at the front of the head's
    optimiser launch, and -- with the rank's rows of the next batch announced to the update -- the gather too. Head and pose-network
    parameters, all moments, the refined poses, the schedule state and the log equal the fused single-GPU steps bit for bit, with wrong and
    empty announcements, fused steps and state reads in between, ragged batch sizes."""
    from tests.helpers import big_problem as _big_problem
    from tests.test_head_gpu import _trainer
    from oracle import head_oracle
    prob = _big_problem(n_images=8, patches_per_view=512)
    flat0 = head_oracle.init_params(helpers.SEED + 1)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_posemlp"], prob)
    cfg.update(global_batch=2048, iterations=60, refine_calibration=True)
    plain, piped = (_trainer(prob, flat0, cfg, max_batch=2048, dtype=dtype) for _ in range(2))
    rng = np.random.default_rng(31)
    N = prob["features"].shape[0]
    batches = [torch.from_numpy(rng.permutation(N)[:(2048 if i % 5 else 1111)].astype(np.int64)).cuda() for i in range(24)]
    other = torch.from_numpy(rng.permutation(N)[:2048].astype(np.int64)).cuda()
    empty = torch.zeros(0, dtype=torch.int64, device="cuda")
    for i, b in enumerate(batches):
        plain.step(b)
        nxt = batches[i + 1] if i + 1 < len(batches) else None
        if i % 7 == 3:
            nxt = other
        if i % 13 == 6:
            nxt = empty
        if i % 11 == 5:
            piped.step(b, nxt)
        else:
            piped.backward(b)
            piped.update(nxt)
        if i % 9 == 4:
            assert plain.state() == piped.state()
            np.testing.assert_array_equal(plain.current_poses(), piped.current_poses())
    torch.cuda.synchronize()
    assert torch.equal(plain.params, piped.params) and torch.equal(plain.adam_m, piped.adam_m) and torch.equal(plain.adam_v, piped.adam_v)
    assert torch.equal(plain.pose_params, piped.pose_params) and torch.equal(plain.pose_m, piped.pose_m) and torch.equal(plain.pose_v, piped.pose_v)
    sp, sq = plain.state(), piped.state()
    assert sp == sq and sp["iteration"] == 24
    lp, lq = plain.log(0, 24), piped.log(0, 24)
    assert np.array_equal(lp[0], lq[0]) and np.array_equal(lp[1], lq[1])
    np.testing.assert_array_equal(plain.current_poses(), piped.current_poses())

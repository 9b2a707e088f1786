"""Training-buffer creation on the GPU against the oracles, through the C ABI."""
import numpy as np
import pytest
import torch

from acezero_amd import synth
from oracle import buffer_oracle as bo
from oracle import encoder_oracle

pytestmark = pytest.mark.gpu


def _views(n, h, w, seed):
    img = torch.from_numpy(synth.make_gray_images(seed=seed, n=n, h=h, w=w))
    rng = np.random.default_rng(seed)
    mask = torch.from_numpy((rng.uniform(size=(n, 1, h, w)) < 0.7).astype(np.float32))
    eye = torch.eye(4).repeat(n, 1, 1)
    K = torch.tensor([[100.0, 0, w / 2], [0, 100.0, h / 2], [0, 0, 1]]).repeat(n, 1, 1)
    return img, mask, eye.clone(), eye.clone(), K, torch.linalg.inv(K)


def test_sampled_rows_match_oracle_bit_for_bit():
    import ctypes as C
    from acezero_amd import _native as N
    from acezero_amd.encoder import Encoder, output_size
    sd = encoder_oracle.init_weights(seed=4099)
    n, h, w, S = 3, 64, 96, 200
    img, mask, *_ = _views(n, h, w, 31)
    enc = Encoder(sd, max_frames=4, max_h=h, max_w=w)
    rows = enc.features_rows(img)
    oh, ow = output_size(h, w)
    m = (torch.nn.functional.interpolate(mask, size=(oh, ow), mode="nearest") > 0).to(torch.uint8).cuda().contiguous()
    of = torch.empty((n * S, 512), dtype=torch.bfloat16, device="cuda")
    op = torch.empty((n * S, 2), dtype=torch.float32, device="cuda")
    ov = torch.empty((n * S,), dtype=torch.int32, device="cuda")
    opix = torch.empty((n * S,), dtype=torch.int32, device="cuda")
    N.check(N.lib().acez_buffer_sample_views(C.c_void_p(rows.data_ptr()), C.c_void_p(m.data_ptr()), n, oh, ow, 512, S, C.c_uint64(2089),
                                             C.c_uint64(10), 7, C.c_void_p(of.data_ptr()), C.c_void_p(op.data_ptr()),
                                             C.c_void_p(ov.data_ptr()), C.c_void_p(opix.data_ptr()), None))
    torch.cuda.synchronize()
    mk = m.cpu().numpy()
    for v in range(n):
        ref = bo.sample_view(mk[v, 0], S, 2089, 10 + v)
        got = opix[v * S:(v + 1) * S].cpu().numpy()
        assert np.array_equal(got, ref)                                              # integer work: bit-exact
        assert np.array_equal(op[v * S:(v + 1) * S].cpu().numpy(), bo.target_px(ref, ow))
        assert bool((ov[v * S:(v + 1) * S] == 7 + v).all())
        src = rows[v * oh * ow + torch.from_numpy(ref).long().cuda()]
        assert torch.equal(of[v * S:(v + 1) * S], src)                               # a row copy


def test_builder_fills_truncates_skips_and_trains():
    from acezero_amd.buffer import BufferBuilder
    from acezero_amd.encoder import Encoder
    from acezero_amd.head import HeadTrainer
    sd = encoder_oracle.init_weights(seed=4099)
    h, w = 64, 96
    enc = Encoder(sd, max_frames=4, max_h=h, max_w=w)
    bld = BufferBuilder(enc, capacity=1000, samples_per_image=256, seed=2089)
    img, mask, aug, pose, K, Kinv = _views(3, h, w, 5)
    mask[1] = 0.0                                   # a view without valid pixels is skipped (ace_trainer.py:377-378)
    took = bld.add_views(img, mask, aug, pose, K, Kinv, [0, 1, 2])
    assert took == 512 and bld.n_views == 2
    took = bld.add_views(img, None, aug, pose, K, Kinv, [0, 1, 2])
    assert took == 488 and bld.full               # truncated at the capacity (ace_trainer.py:415-416)
    assert bld.add_views(img, None, aug, pose, K, Kinv, [0, 1, 2]) == 0
    buf = bld.finish()
    assert buf["features"].shape == (1000, 512) and buf["view_aug_inv"].shape[1:] == (3, 4)
    assert int(buf["view_idx"].max()) == buf["view_aug_inv"].shape[0] - 1
    tr = HeadTrainer(torch.zeros(3), max_batch=512, iterations=10, schedule="circle")
    tr.set_buffer(**buf)
    idx = torch.randperm(1000, device="cuda")[:512]
    tr.step(idx)
    st = tr.state()
    assert st["iteration"] == 1 and np.isfinite(st["loss"])


@pytest.mark.parametrize("jitter", [False, True])
@pytest.mark.parametrize("scale", [0.94, 1.0, 1.06])
def test_device_warp_equals_the_framework_warp(scale, jitter):
    """acez_buffer_warp_views (one launch: warp + ColorJitter + the mask at feature resolution) against session.warp_views, the torch
    restatement of dataset.py:283-343 the CPU tests pin (affine_grid / grid_sample with reflection padding, a ones image through the same
    warp with zero padding, F.interpolate(nearest) to the feature map, ace_trainer.py:373-374)."""
    import torch.nn.functional as F
    from acezero_amd import session
    from acezero_amd.encoder import output_size
    rng = np.random.default_rng(11)
    imgs = torch.from_numpy(synth.make_gray_images(seed=3, n=5, h=120, w=168)).cuda().contiguous()
    B = 7
    idx = rng.integers(0, 5, size=B)
    ang = np.radians(rng.uniform(-15, 15, size=B))
    jit = (rng.uniform(0.8, 1.2, size=B), rng.uniform(0.8, 1.2, size=B)) if jitter else None
    ref_v, ref_m, _ = session.warp_views(imgs[torch.from_numpy(idx).cuda()], scale, ang, jit)
    hs, ws = ref_v.shape[-2:]
    oh, ow = output_size(hs, ws)
    views, mask = session.warp_views_device(imgs, idx, scale, ang, jit, mask_hw=(oh, ow))
    assert views.shape == ref_v.shape and mask.shape == (B, 1, oh, ow) and mask.dtype == torch.uint8
    # same formulas in fp32; the framework evaluates the affine map as a batched matrix product, here it is two fused multiply-adds per
    # coordinate: source coordinates agree to ~1e-4 pixel, values to that times the local gradient
    d = (views - ref_v).abs()
    assert float(d.max()) < 2e-3 * float(ref_v.abs().max()) and float(d.mean()) < 2e-5 * float(ref_v.abs().max()), (float(d.max()), float(d.mean()))
    ref_mask = F.interpolate(ref_m.float(), size=(oh, ow), mode="nearest") > 0
    differ = int((ref_mask != (mask > 0)).sum())
    assert differ <= 2, differ            # a cell exactly on the frame's edge may fall either way
    assert 0.3 < float((mask > 0).float().mean()) < 1.0

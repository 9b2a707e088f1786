"""Registration pipeline on the GPU: images -> encoder -> head -> RANSAC without leaving HBM."""
import numpy as np
import pytest
import torch

from acezero_amd import synth
from oracle import encoder_oracle, head_oracle

pytestmark = pytest.mark.gpu


def _head_state_dict(seed=3):
    from acezero_amd.head import layer_names
    flat = head_oracle.init_params(seed, 1, True)
    sd, o = {}, 0
    for name in layer_names(1):
        sd[name + ".weight"] = flat[o:o + 262144].view(512, 512, 1, 1).clone(); o += 262144
        sd[name + ".bias"] = flat[o:o + 512].clone(); o += 512
    sd["fc3.weight"] = flat[o:o + 2048].view(4, 512, 1, 1).clone(); o += 2048
    sd["fc3.bias"] = flat[o:o + 4].clone()
    sd["mean"] = torch.tensor([1.0, -2.0, 0.5]).view(1, 3, 1, 1)
    return sd, flat


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_regressor_forward_matches_oracles(dtype):
    from acezero_amd.network import Regressor
    esd = encoder_oracle.init_weights(seed=4099)
    hsd, flat = _head_state_dict()
    img = torch.from_numpy(synth.make_gray_images(seed=21, n=3, h=96, w=128))
    net = Regressor.create_from_split_state_dict(esd, hsd, max_frames=2, max_h=96, max_w=128, dtype=dtype)
    sc = net(img)
    assert sc.shape == (3, 3, 12, 16) and sc.dtype == torch.float32 and sc.is_cuda
    mean = torch.tensor([1.0, -2.0, 0.5])
    # rounding-matched oracle: 16-bit encoder -> 16-bit head forward
    rows = encoder_oracle.EncoderOracle(esd, dtype).features_rows(img)
    ref = head_oracle.HeadOracle(flat, mean, 1, True, mode=dtype).scene_coordinates(rows).view(3, 12, 16, 3).permute(0, 3, 1, 2)
    err = (sc.cpu() - ref).abs().max().item()
    scale = (ref - mean.view(1, 3, 1, 1)).abs().max().item()
    assert err < (2e-2 if dtype == "bf16" else 2e-3) * scale, (err, scale)
    if dtype == "fp16":
        # the reference's arithmetic against the UN-ROUNDED fp32 oracles (both pinned on the reference's own modules): images -> scene
        # coordinates within 2e-3 of the coordinate scale (bf16: 2e-2)
        rows32 = encoder_oracle.EncoderOracle(esd, "fp32").features_rows(img)
        ref32 = head_oracle.HeadOracle(flat, mean, 1, True, mode="fp32").scene_coordinates(rows32).view(3, 12, 16, 3).permute(0, 3, 1, 2)
        err32 = (sc.cpu() - ref32).abs().max().item()
        assert err32 < 2e-3 * scale, (err32, scale)
    # the two-step path (features as a tensor, then the head) gives the same maps as the fused row path
    sc2 = net.get_scene_coordinates(net.get_features(img))
    assert torch.equal(sc2, sc)


def test_register_images_equals_stagewise_composition():
    from acezero_amd import dsacstar
    from acezero_amd.network import Regressor
    esd = encoder_oracle.init_weights(seed=4099)
    hsd, _ = _head_state_dict()
    img = torch.from_numpy(synth.make_gray_images(seed=4, n=2, h=96, w=128))
    net = Regressor.create_from_split_state_dict(esd, hsd, max_frames=2, max_h=96, max_w=128)
    params = dict(hyps=16, thr=10.0, alpha=100.0, max_reproj=100.0, sub=8, max_tries=16)
    intr = [(100.0, 64.0, 48.0)] * 2
    poses, inl, _ = net.register(img, intr, params, seed=11, frame_ids=[5, 6])
    sc = net(img)
    poses2, inl2, _ = dsacstar.register_batch(sc, intr, params, 11, [5, 6], want_masks=False)
    assert torch.equal(poses, poses2) and torch.equal(inl, inl2)
    assert poses.shape == (2, 4, 4) and bool(torch.isfinite(poses).all())


def test_create_from_encoder_and_load_encoder(tmp_path):
    """Regressor.create_from_encoder (ace_network.py:177-199) and load_encoder (:253-257)."""
    from acezero_amd.network import Regressor
    esd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights(seed=4099).items()}
    net = Regressor.create_from_encoder(esd, mean=torch.tensor([1.0, 2.0, 3.0]), num_head_blocks=2, use_homogeneous=False, seed=3, max_frames=2,
                                        max_h=64, max_w=96)
    sd = net.heads.state_dict()
    assert "1c2.weight" in sd and "2c0.weight" not in sd and sd["fc3.weight"].shape == (3, 512, 1, 1)
    assert float(sd["fc1.weight"].abs().max()) <= 1.0 / 512 ** 0.5 + 1e-7 and torch.equal(sd["mean"].view(3).cpu(), torch.tensor([1.0, 2.0, 3.0]))
    img = torch.from_numpy(synth.make_gray_images(seed=2, n=2, h=64, w=96))
    a = net(img).clone()
    assert a.shape == (2, 3, 8, 12) and bool(torch.isfinite(a).all())
    esd2 = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights(seed=5).items()}
    torch.save(esd2, tmp_path / "enc.pt")
    net.load_encoder(tmp_path / "enc.pt")
    b = net(img)
    assert not torch.allclose(a, b)
    ref = Regressor.create_from_split_state_dict(esd2, {k: v.cpu() for k, v in sd.items()}, max_frames=2, max_h=64, max_w=96)(img)
    assert torch.equal(b, ref)

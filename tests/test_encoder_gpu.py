"""GPU parity of the HIP encoder (through the C ABI) against oracle/encoder_oracle.py in its rounding-matched modes (bf16, fp16) and
against the reference Encoder's own outputs (tests/golden/encoder_small.npz: fp32, and fp16 autocast)."""
import numpy as np
import pytest
import torch

from acezero_amd import synth
from oracle import encoder_oracle

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


# (rounding-matched oracle: relative L2, max abs / max |ref|; un-rounded fp32 arithmetic: relative L2). One last-place flip of a 16-bit
# rounding is 2^-8 (bf16) / 2^-11 (fp16) relative; the following layers average the flips out.
TOL = {"bf16": (4e-3, 0.03, 2e-2), "fp16": (5e-4, 4e-3, 2e-3)}


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("tile", ["0", "80", "256", "512", "3"])   # 3: the 3x3 patch kernel where the layer allows it
@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 120, 200), (3, 41, 77)])
def test_encoder_matches_rounding_matched_oracle(shape, tile, dtype, monkeypatch, diag_lib):   # ACEZ_CONV_TILE: diagnostics build
    from acezero_amd.encoder import Encoder, output_size
    monkeypatch.setenv("ACEZ_CONV_TILE", tile)   # 256: the large-M kernel on small inputs (ragged last tiles everywhere)
    n, h, w = shape
    sd = encoder_oracle.init_weights(seed=4099)
    img = torch.from_numpy(synth.make_gray_images(seed=5 + h, n=n, h=h, w=w))
    orc = encoder_oracle.EncoderOracle(sd, dtype)
    ref = orc.forward(img)                       # [n, 512, oh, ow]
    enc = Encoder(sd, max_frames=2, max_h=h, max_w=w, dtype=dtype)   # max_frames < n exercises the chunking
    out = enc(img).cpu()
    oh, ow = output_size(h, w)
    assert out.shape == ref.shape == (n, 512, oh, ow)
    # 16-bit operands, fp32 accumulation on both sides; the summation order differs and single roundings of intermediate activations can
    # flip by one ulp
    rel, mx, rel32 = TOL[dtype]
    assert _rel(out, ref) < rel, _rel(out, ref)
    assert float((out - ref).abs().max()) < mx * float(ref.abs().max())
    # and against the un-rounded reference arithmetic
    ref32 = encoder_oracle.EncoderOracle(sd, "fp32").forward(img)
    assert _rel(out, ref32) < rel32, _rel(out, ref32)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_encoder_golden_reference_features(dtype):
    """Against the reference's own Encoder outputs (tests/golden/encoder_small.npz): its fp32 forward -- bf16 within 2e-2, fp16 within
    2e-3 (north_star's 1e-3 is a bound on scene coordinates; the reference's own fp16 autocast is 4e-4 from its fp32) -- and, in fp16, its
    forward under torch.autocast(float16), the arithmetic the reference actually runs."""
    import os
    from acezero_amd.encoder import Encoder
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_small.npz"))
    sd = encoder_oracle.init_weights(seed=4099)
    img = torch.from_numpy(synth.make_gray_images(seed=77, n=2, h=64, w=96))
    out = Encoder(sd, max_frames=4, max_h=64, max_w=96, dtype=dtype)(img).cpu()
    ref = torch.from_numpy(g["features"])
    assert _rel(out, ref) < TOL[dtype][2], _rel(out, ref)
    if dtype == "fp16":
        ref16 = torch.from_numpy(g["features_fp16_autocast"].astype(np.float32))
        assert _rel(out, ref16) < 5e-4, _rel(out, ref16)
        assert float((out - ref16).abs().max()) < 4e-3 * float(ref16.abs().max())


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_encoder_rows_feed_the_head_layout(dtype):
    from acezero_amd.encoder import Encoder
    sd = encoder_oracle.init_weights(seed=4099)
    img = torch.from_numpy(synth.make_gray_images(seed=9, n=2, h=64, w=96))
    enc = Encoder(sd, max_frames=2, max_h=64, max_w=96, dtype=dtype)
    rows = enc.features_rows(img)
    assert rows.dtype == (torch.float16 if dtype == "fp16" else torch.bfloat16) and rows.shape == (2 * 8 * 12, 512)
    f = enc(img)
    assert torch.equal(rows[1 * 96 + 3 * 12 + 5].float().cpu(), f[1, :, 3, 5].cpu())


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("hw", [(480, 640), (480, 741)])   # 7-Scenes frames; Mip-NeRF 360 garden at images_4 (BASELINE configs 1-3)
def test_encoder_and_regressor_at_baseline_frame_sizes(hw, dtype):
    """The sizes at which conv12p / conv3x3r / convgemm512 pick their real tilings and chunking: encoder features and the
    scene-coordinate maps of Regressor.forward against the rounding-matched oracles; in fp16 (the reference's arithmetic) the maps also
    against the UN-ROUNDED fp32 oracles at 2e-3 of the coordinate scale."""
    from acezero_amd.encoder import Encoder, output_size
    from acezero_amd.network import Regressor
    from oracle import head_oracle
    from tests.test_pipeline_gpu import _head_state_dict
    h, w = hw
    n = 3
    sd = encoder_oracle.init_weights(seed=4099)
    img = torch.from_numpy(synth.make_gray_images(seed=31 + w, n=n, h=h, w=w))
    orc = encoder_oracle.EncoderOracle(sd, dtype)
    ref = orc.forward(img)
    oh, ow = output_size(h, w)
    out = Encoder(sd, max_frames=n, max_h=h, max_w=w, dtype=dtype)(img).cpu()
    assert out.shape == ref.shape == (n, 512, oh, ow) and (oh, ow) == (60, (w + 7) // 8)
    assert _rel(out, ref) < TOL[dtype][0], _rel(out, ref)
    assert float((out - ref).abs().max()) < TOL[dtype][1] * float(ref.abs().max())
    hsd, flat = _head_state_dict()
    net = Regressor.create_from_split_state_dict(sd, hsd, max_frames=n, max_h=h, max_w=w, dtype=dtype)
    sc = net(img).cpu()
    rows = orc.features_rows(img)
    mean = torch.tensor([1.0, -2.0, 0.5])
    Xo = head_oracle.HeadOracle(flat, mean, 1, True, mode=dtype).scene_coordinates(rows).view(n, oh, ow, 3).permute(0, 3, 1, 2)
    err = (sc - Xo).abs().max().item()
    scale = (Xo - mean.view(1, 3, 1, 1)).abs().max().item()
    assert err < (2e-2 if dtype == "bf16" else 2e-3) * scale, (err, scale)
    if dtype == "fp16":
        rows32 = encoder_oracle.EncoderOracle(sd, "fp32").features_rows(img)
        X32 = head_oracle.HeadOracle(flat, mean, 1, True, mode="fp32").scene_coordinates(rows32).view(n, oh, ow, 3).permute(0, 3, 1, 2)
        err32 = (sc - X32).abs().max().item()
        assert err32 < 2e-3 * scale, (err32, scale)

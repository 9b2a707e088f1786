"""GPU parity of the HIP encoder (through the C ABI) against oracle/encoder_oracle.py in bf16 mode."""
import numpy as np
import pytest
import torch

from acezero_amd import synth
from oracle import encoder_oracle

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("tile", ["0", "80", "256", "512", "3"])   # 3: the 3x3 patch kernel where the layer allows it
@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 120, 200), (3, 41, 77)])
def test_encoder_matches_bf16_oracle(shape, tile, monkeypatch, diag_lib):   # ACEZ_CONV_TILE / ACEZ_CONV12: diagnostics build
    from acezero_amd.encoder import Encoder, output_size
    monkeypatch.setenv("ACEZ_CONV_TILE", tile)   # 256: the large-M kernel on small inputs (ragged last tiles everywhere)
    n, h, w = shape
    sd = encoder_oracle.init_weights(seed=4099)
    img = torch.from_numpy(synth.make_gray_images(seed=5 + h, n=n, h=h, w=w))
    orc = encoder_oracle.EncoderOracle(sd, "bf16")
    ref = orc.forward(img)                       # [n, 512, oh, ow]
    enc = Encoder(sd, max_frames=2, max_h=h, max_w=w)   # max_frames < n exercises the chunking
    out = enc(img).cpu()
    oh, ow = output_size(h, w)
    assert out.shape == ref.shape == (n, 512, oh, ow)
    # bf16 operands, fp32 accumulation on both sides; the summation order differs and single bf16 roundings of
    # intermediate activations can flip by one ulp (2^-8 relative), which the following layers average out
    assert _rel(out, ref) < 4e-3, _rel(out, ref)
    assert float((out - ref).abs().max()) < 0.03 * float(ref.abs().max())
    # and against the un-rounded reference arithmetic: bf16-level agreement
    ref32 = encoder_oracle.EncoderOracle(sd, "fp32").forward(img)
    assert _rel(out, ref32) < 2e-2


def test_separate_conv1_conv2_path_still_matches(monkeypatch, diag_lib):
    from acezero_amd.encoder import Encoder
    monkeypatch.setenv("ACEZ_CONV12", "0")
    sd = encoder_oracle.init_weights(seed=4099)
    img = torch.from_numpy(synth.make_gray_images(seed=3, n=2, h=72, w=100))
    ref = encoder_oracle.EncoderOracle(sd, "bf16").forward(img)
    out = Encoder(sd, max_frames=2, max_h=72, max_w=100)(img).cpu()
    assert _rel(out, ref) < 4e-3


@pytest.mark.parametrize("shape", [(2, 72, 100), (3, 41, 77), (2, 480, 640)])
def test_pipelined_conv12_equals_the_kernel_it_replaced_bitwise(shape, monkeypatch, diag_lib):
    """conv12p_kernel (round 5: conv1 of tile j + 1 on four waves under conv2 of tile j on the other four, image patches staged two tiles
    ahead) against conv12_kernel (ACEZ_CONV12P=0, diagnostics build): same products, same accumulation order, same roundings -- the
    encoder's output must agree bit for bit, at ragged sizes (partial tiles, a workgroup with a single tile) and at 7-Scenes frames."""
    from acezero_amd.encoder import Encoder
    n, h, w = shape
    sd = encoder_oracle.init_weights(seed=4099)
    img = torch.from_numpy(synth.make_gray_images(seed=11 + w, n=n, h=h, w=w))
    new = Encoder(sd, max_frames=n, max_h=h, max_w=w)(img).cpu()
    monkeypatch.setenv("ACEZ_CONV12P", "0")
    old = Encoder(sd, max_frames=n, max_h=h, max_w=w)(img).cpu()
    assert torch.equal(new, old)


def test_encoder_golden_reference_features():
    """Against the reference's own Encoder output (tests/golden/encoder_small.npz), at bf16 accuracy."""
    import os
    from acezero_amd.encoder import Encoder
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_small.npz"))
    sd = encoder_oracle.init_weights(seed=4099)
    img = torch.from_numpy(synth.make_gray_images(seed=77, n=2, h=64, w=96))
    out = Encoder(sd, max_frames=4, max_h=64, max_w=96)(img).cpu()
    ref = torch.from_numpy(g["features"])
    assert _rel(out, ref) < 2e-2


def test_encoder_rows_feed_the_head_layout():
    from acezero_amd.encoder import Encoder
    sd = encoder_oracle.init_weights(seed=4099)
    img = torch.from_numpy(synth.make_gray_images(seed=9, n=2, h=64, w=96))
    enc = Encoder(sd, max_frames=2, max_h=64, max_w=96)
    rows = enc.features_rows(img)
    assert rows.dtype == torch.bfloat16 and rows.shape == (2 * 8 * 12, 512)
    f = enc(img)
    assert torch.equal(rows[1 * 96 + 3 * 12 + 5].float().cpu(), f[1, :, 3, 5].cpu())


@pytest.mark.parametrize("hw", [(480, 640), (480, 741)])   # 7-Scenes frames; Mip-NeRF 360 garden at images_4 (BASELINE configs 1-3)
def test_encoder_and_regressor_at_baseline_frame_sizes(hw):
    """The sizes at which conv12 / conv3x3p / convgemm512 pick their real tilings and chunking (only toy frames were
    compared with the oracle): encoder features and the scene-coordinate maps of Regressor.forward against the bf16 oracles."""
    from acezero_amd.encoder import Encoder, output_size
    from acezero_amd.network import Regressor
    from oracle import head_oracle
    from tests.test_pipeline_gpu import _head_state_dict
    h, w = hw
    n = 3
    sd = encoder_oracle.init_weights(seed=4099)
    img = torch.from_numpy(synth.make_gray_images(seed=31 + w, n=n, h=h, w=w))
    orc = encoder_oracle.EncoderOracle(sd, "bf16")
    ref = orc.forward(img)
    oh, ow = output_size(h, w)
    out = Encoder(sd, max_frames=n, max_h=h, max_w=w)(img).cpu()
    assert out.shape == ref.shape == (n, 512, oh, ow) and (oh, ow) == (60, (w + 7) // 8)
    assert _rel(out, ref) < 4e-3, _rel(out, ref)
    assert float((out - ref).abs().max()) < 0.03 * float(ref.abs().max())
    hsd, flat = _head_state_dict()
    net = Regressor.create_from_split_state_dict(sd, hsd, max_frames=n, max_h=h, max_w=w)
    sc = net(img).cpu()
    rows = orc.features_rows(img)
    mean = torch.tensor([1.0, -2.0, 0.5])
    Xo = head_oracle.HeadOracle(flat, mean, 1, True, mode="bf16").scene_coordinates(rows).view(n, oh, ow, 3).permute(0, 3, 1, 2)
    err = (sc - Xo).abs().max().item()
    scale = (Xo - mean.view(1, 3, 1, 1)).abs().max().item()
    assert err < 2e-2 * scale, (err, scale)

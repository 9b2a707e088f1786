"""The encoder oracle against the reference's own Encoder (golden produced by tests/golden/make_encoder_golden.py)."""
import os

import numpy as np
import torch

from acezero_amd import synth
from oracle import encoder_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "encoder_small.npz")


def _inputs():
    sd = encoder_oracle.init_weights(seed=4099)
    img = torch.from_numpy(synth.make_gray_images(seed=77, n=2, h=64, w=96))
    return sd, img


def test_fp32_oracle_matches_reference_encoder():
    g = np.load(GOLD)
    sd, img = _inputs()
    # the fixture was generated from the same seeded inputs
    assert abs(float(img.double().sum()) - float(g["image_sum"])) < 1e-6
    assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(g["weight_sum"])) < 1e-6
    out = encoder_oracle.EncoderOracle(sd, "fp32").forward(img).numpy()
    ref = g["features"]
    assert out.shape == ref.shape == (2, 512, 8, 12)
    # same arithmetic (torch conv2d fp32), possibly a different summation order inside the BLAS/oneDNN kernels
    assert np.max(np.abs(out - ref)) < 2e-6 * max(1.0, float(np.max(np.abs(ref))))


def test_bf16_mode_stays_close_to_fp32():
    sd, img = _inputs()
    a = encoder_oracle.EncoderOracle(sd, "fp32").forward(img)
    b = encoder_oracle.EncoderOracle(sd, "bf16").forward(img)
    rel = float((a - b).norm() / a.norm())
    assert rel < 2e-2, rel
    # values of the bf16 mode are exactly representable in bf16
    assert torch.equal(b, b.to(torch.bfloat16).to(torch.float32))


def test_fp16_mode_matches_reference_encoder_under_fp16_autocast():
    """The fp16 mode restates what autocast does to the reference Encoder; the golden is the reference itself under
    torch.autocast("cpu", float16). oneDNN's accumulation order differs from F.conv2d's on fp32 copies of the half operands, so
    last-place flips of half roundings remain: agreement far inside one half ulp of the feature scale on average, a few ulps at most."""
    g = np.load(GOLD)
    sd, img = _inputs()
    out = encoder_oracle.EncoderOracle(sd, "fp16").forward(img)
    assert torch.equal(out, out.to(torch.float16).to(torch.float32))        # half-representable values
    ref = torch.from_numpy(g["features_fp16_autocast"].astype(np.float32))
    rel = float((out - ref).norm() / ref.norm())
    assert rel < 4e-4, rel
    assert float((out - ref).abs().max()) < 4e-3 * float(ref.abs().max())
    # and the reference's own two precisions differ by more than the oracle differs from its autocast mode
    ref32 = torch.from_numpy(g["features"])
    assert rel < float((ref - ref32).norm() / ref32.norm())


def test_rows_layout_is_pixel_major():
    sd, img = _inputs()
    o = encoder_oracle.EncoderOracle(sd, "fp32")
    f = o.forward(img)
    rows = o.features_rows(img)
    assert rows.shape == (2 * 8 * 12, 512)
    assert torch.equal(rows[1 * 96 + 3 * 12 + 5], f[1, :, 3, 5])

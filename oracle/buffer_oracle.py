"""CPU oracle of the training-buffer sampling -- TEST INFRASTRUCTURE ONLY (see oracle/head_oracle.py for the rule).

Restates ace_trainer.py:373-431 for one batch of views: nearest-neighbour mask at feature resolution, `samples` draws per
view uniformly with replacement among the valid pixels, target pixel 8 * (x + 0.5, y + 0.5) (ace_util.py:7-13).
torch.multinomial's random stream is not reproducible outside torch, so -- exactly as for the RANSAC path -- parity is
defined on a counter-based stream keyed by (seed, view id, sample) shared by this oracle and the HIP kernel; the
DISTRIBUTION (uniform over valid pixels, with replacement) is the reference's and is what tests/test_buffer_oracle.py checks.
"""
import numpy as np

M64 = (1 << 64) - 1


def smix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def sample_draw(seed, view_id, s):
    return smix64((smix64((seed ^ ((view_id * 0xD1342543DE82EF95) & M64)) & M64) + s) & M64) >> 32


def sample_view(mask_hw, samples, seed, view_id):
    """mask_hw: bool/uint8 [h, w] (or None). Returns int32 [samples] of pixel indices y*w+x."""
    valid = np.flatnonzero(np.asarray(mask_hw).reshape(-1) != 0)
    n = len(valid)
    assert n > 0
    out = np.zeros(samples, np.int32)
    for s in range(samples):
        k = (sample_draw(seed, view_id, s) * n) >> 32
        out[s] = valid[k]
    return out


def target_px(pix, w):
    y, x = np.divmod(pix, w)
    return np.stack([8.0 * (x + 0.5), 8.0 * (y + 0.5)], axis=1).astype(np.float32)

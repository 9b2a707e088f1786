"""Properties of the training-buffer sampling oracle (ace_trainer.py:404-431 restated in oracle/buffer_oracle.py)."""
import numpy as np

from oracle import buffer_oracle as bo


def test_draws_are_uniform_over_valid_pixels_with_replacement():
    rng = np.random.default_rng(0)
    mask = rng.uniform(size=(12, 16)) < 0.4
    mask[0, 0] = True
    pix = bo.sample_view(mask, 20000, seed=2089, view_id=3)
    valid = np.flatnonzero(mask.reshape(-1))
    assert set(np.unique(pix)) <= set(valid)                 # only valid pixels
    counts = np.array([(pix == v).sum() for v in valid])
    assert counts.min() > 0                                   # all of them reachable
    exp = 20000 / len(valid)
    chi2 = float(((counts - exp) ** 2 / exp).sum())
    assert chi2 < 2.0 * len(valid)                            # ~ chi-square with len(valid)-1 dof
    assert len(np.unique(pix)) < len(pix)                     # with replacement


def test_stream_is_keyed_by_view_and_sample_not_by_batching():
    mask = np.ones((6, 8), bool)
    a = bo.sample_view(mask, 64, 7, view_id=5)
    b = bo.sample_view(mask, 32, 7, view_id=5)
    assert np.array_equal(a[:32], b)
    assert not np.array_equal(a, bo.sample_view(mask, 64, 7, view_id=6))
    assert not np.array_equal(a, bo.sample_view(mask, 64, 8, view_id=5))


def test_target_pixels_are_cell_centres():
    px = bo.target_px(np.array([0, 1, 8, 17]), w=8)
    assert np.array_equal(px, np.array([[4, 4], [12, 4], [4, 12], [12, 20]], np.float32))   # ace_util.py:7-13

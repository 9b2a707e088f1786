"""Checkpoint compatibility: the key names / shapes this package reads and writes are the reference Regressor's
(tests/golden/state_dict_keys.json, captured from ace_network.Regressor by tests/golden/make_state_dict_golden.py)."""
import json
import os

import numpy as np
import pytest

from acezero_amd import synth

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_keys.json")))


@pytest.mark.parametrize("blocks,homog", [(1, True), (0, False), (2, True)])
def test_head_and_encoder_keys_match_reference(blocks, homog):
    ref = GOLD[f"blocks{blocks}_homog{int(homog)}"]
    enc = {"encoder." + k: list(v.shape) for k, v in synth.init_encoder_weights().items()}
    flat = synth.init_head_params(1, blocks, homog)
    head = {"heads." + k: list(np.asarray(v).shape) for k, v in synth.head_state_dict(flat, blocks, homog).items()}
    ours = dict(enc)
    ours.update(head)
    # buffers of the homogeneous head (ace_network.py:109-115) are constants derived from two scalars: not parameters
    const = {"heads.max_scale", "heads.min_scale", "heads.max_inv_scale", "heads.h_beta", "heads.min_inv_scale"}
    assert set(ours) == set(ref) - const, set(ours) ^ (set(ref) - const)
    for k, shp in ours.items():
        assert shp == ref[k], (k, shp, ref[k])
    assert flat.size == synth.head_num_params(blocks, homog)


def test_encoder_layer_order_is_the_c_abi_order():
    from acezero_amd.encoder import LAYER_NAMES
    assert [n for n, *_ in synth.ENCODER_LAYERS] == LAYER_NAMES

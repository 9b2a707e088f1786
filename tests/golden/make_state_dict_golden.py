"""Capture the reference Regressor's state_dict keys and shapes (tests/golden/state_dict_keys.json).

Run in the build container only (needs /root/reference):   python tests/golden/make_state_dict_golden.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, "/root/reference")
import ace_network  # noqa: E402

out = {}
for blocks, homog in ((1, True), (0, False), (2, True)):
    net = ace_network.Regressor(torch.zeros(3), blocks, homog)
    out[f"blocks{blocks}_homog{int(homog)}"] = {k: list(v.shape) for k, v in net.state_dict().items()}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "state_dict_keys.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path, {k: len(v) for k, v in out.items()})

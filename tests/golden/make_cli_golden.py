"""Dump the reference's argparse surface (flag names, defaults, choices) of train_ace.py, register_mapping.py, ace_zero.py and export_point_cloud.py
to tests/golden/cli_flags.json. Build container only (needs /root/reference). The parsers are captured by running
the scripts with argparse.ArgumentParser.parse_args patched to raise after construction."""
import argparse
import json
import os
import runpy
import sys
from pathlib import Path
from unittest.mock import MagicMock

REF = "/root/reference"
for name in ["torchvision", "torchvision.transforms", "torchvision.transforms.functional", "skimage", "skimage.transform", "skimage.io",
             "skimage.color", "skimage.draw", "roma", "pyrender", "trimesh", "matplotlib.pyplot", "cv2", "dsacstar"]:
    sys.modules.setdefault(name, MagicMock())
sys.path.insert(0, REF)


class Captured(Exception):
    pass


def capture(script):
    box = {}

    def fake_parse(self, *a, **k):
        box["p"] = self
        raise Captured()
    orig = argparse.ArgumentParser.parse_args
    argparse.ArgumentParser.parse_args = fake_parse
    try:
        runpy.run_path(os.path.join(REF, script), run_name="__main__")
    except Captured:
        pass
    finally:
        argparse.ArgumentParser.parse_args = orig
    out = {}
    for act in box["p"]._actions:
        if act.dest == "help":
            continue
        d = act.default
        if isinstance(d, Path):
            d = "<path>" if "ace_encoder_pretrained" in str(d) else str(d)
        out[act.dest] = {"flags": list(act.option_strings), "default": d, "choices": list(act.choices) if act.choices else None,
                         "positional": not act.option_strings}
    return out


if __name__ == "__main__":
    for name in ["joblib", "dataset_io"]:
        sys.modules.setdefault(name, MagicMock())
    res = {"train_ace": capture("train_ace.py"), "register_mapping": capture("register_mapping.py"), "ace_zero": capture("ace_zero.py"),
           "export_point_cloud": capture("export_point_cloud.py")}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cli_flags.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print({k: len(v) for k, v in res.items()})

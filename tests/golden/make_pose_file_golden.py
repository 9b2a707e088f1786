"""Write tests/golden/pose_file_ref.txt with the REFERENCE's dataset_io.write_pose_to_pose_file and parse it back with its
dataset_io.load_dataset_ace -> tests/golden/pose_file_ref.npz (what the reference reads out of that file at threshold 500).
Build container only (needs /root/reference).  torchvision (imported by dataset_io, unused on this path) is stubbed."""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
from scipy.spatial.transform import Rotation

for name in ["torchvision", "torchvision.transforms", "torchvision.transforms.functional"]:
    sys.modules.setdefault(name, MagicMock())
sys.path.insert(0, "/root/reference")
import dataset_io  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.helpers import pose_file_cases as poses  # noqa: E402


if __name__ == "__main__":
    P, conf = poses()
    path = os.path.join(HERE, "pose_file_ref.txt")
    with open(path, "w") as f:
        for i in range(len(P)):
            dataset_io.write_pose_to_pose_file(f, f"scene/frame_{i:03d}.png", P[i], int(conf[i]) if i else float("inf"), 525.0 + i)
    files, c2w, focals = dataset_io.load_dataset_ace(path, 500)
    np.savez(os.path.join(HERE, "pose_file_ref.npz"), files=np.array(files), c2w=np.stack([p.numpy() for p in c2w]), focals=np.array(focals))
    print(len(files), "entries kept of", len(P))

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


def poses(n=12, seed=5):
    """world -> camera matrices (float64) and confidences, seeded."""
    rng = np.random.default_rng(seed)
    out = np.tile(np.eye(4), (n, 1, 1))
    out[:, :3, :3] = Rotation.from_rotvec(rng.normal(0, 1.0, size=(n, 3))).as_matrix()
    out[:, :3, 3] = rng.normal(0, 3.0, size=(n, 3))
    conf = rng.integers(0, 3000, size=n)
    conf[1], conf[2] = 499, 500          # the threshold itself is kept (confidence < threshold is dropped), 499 is not
    return out, conf


if __name__ == "__main__":
    P, conf = poses()
    path = os.path.join(HERE, "pose_file_ref.txt")
    with open(path, "w") as f:
        for i in range(len(P)):
            dataset_io.write_pose_to_pose_file(f, f"scene/frame_{i:03d}.png", P[i], int(conf[i]) if i else float("inf"), 525.0 + i)
    files, c2w, focals = dataset_io.load_dataset_ace(path, 500)
    np.savez(os.path.join(HERE, "pose_file_ref.npz"), files=np.array(files), c2w=np.stack([p.numpy() for p in c2w]), focals=np.array(focals))
    print(len(files), "entries kept of", len(P))

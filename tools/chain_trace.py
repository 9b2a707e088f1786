"""Timeline of one chain_kernel launch from in-kernel s_memtime stamps (workgroup 0: multiplier wave 0 and loader 0).
ACEZ_CHAIN_DBG=<flags> python tools/chain_trace.py"""
import os as _os
_os.environ.setdefault("ACEZ_LIB", "diag")   # the ACEZ_* ablation switches exist in the diagnostics build only (acezero_amd/build.py --diag)
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ACEZ_CHAIN_TRACE"] = "1"
os.environ["ACEZ_CHAIN"] = "1"
import bench  # noqa: E402
from acezero_amd import _native as N, synth  # noqa: E402
from acezero_amd.head import HeadTrainer, _stream  # noqa: E402

dev = torch.device("cuda", 0)
patches = 1_000_000
prob, feats, target_px, view_idx = bench.make_buffer(patches, dev, 2089)
tr = HeadTrainer(prob["mean"], max_batch=5120, loss_type="tanh", schedule="1cyclepoly", iterations=25000, lr_min=0.0005, lr_max=0.003,
                 warmup_iterations=1000, warmup_lr=0.0005, cooldown_iterations=5000)
tr.load_flat(torch.from_numpy(synth.init_head_params(1)))
tr.set_buffer(feats, target_px, view_idx, prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"], prob["view_image"], prob["image_pose_inv"])
perm = torch.randperm(patches, device=dev)
for i in range(20):
    tr.step(perm[i * 5120:(i + 1) * 5120].contiguous())
torch.cuda.synchronize()
out = np.zeros(512, np.uint64)
N.check(tr.lib.acez_trainer_debug_read(tr._h, 5, 0, out.ctypes.data_as(C.c_void_p), out.nbytes, _stream()))
m = out[:256].astype(np.int64)
ld = out[256:].astype(np.int64)
t0 = m[0]
print("dbg", os.environ.get("ACEZ_CHAIN_DBG", "0"), "ticks are s_memtime units (100 MHz = 10 ns each if constant clock; else shader cycles)")
names = ["Xready", "fetch", "Kdone", "alldone", "epi"]
k = 1
for step in range(15):
    row = m[k:k + 5] - t0
    if m[k] == 0:
        break
    print("step %2d: " % step + "  ".join("%s %6d" % (nm, v) for nm, v in zip(names, row)) + "   | K loop %5d  wait-all %5d  epilogue %5d  next-X-wait %5d" % (
        row[2] - row[1], row[3] - row[2], row[4] - row[3], (m[k + 5] - t0 - row[4]) if m[k + 5] else 0))
    k += 5
fine = out[384:448].astype(np.int64)
if fine[0]:
    f = fine[fine > 0].reshape(-1, 4)
    print("step 1, own stages of multiplier 0: [poll start, ready seen, mfma issued, prog set] relative to the first")
    for row in f[:16]:
        print("   ", list(row - f[0, 0]), " wait %5d  mfma %5d  set %5d" % (row[1] - row[0], row[2] - row[1], row[3] - row[2]))
ld = out[256:384].astype(np.int64)
lds = ld[ld > 0] - t0
print("loader 0 signals (first 24):", list(lds[:24]))
print("loader 0 per-layer span:", [int(lds[min(len(lds) - 1, 8 * i + 7)] - lds[8 * i]) for i in range(len(lds) // 8)])

"""Timeline of wgrad_opt_kernel's epilogue (diagnostics build, ACEZ_WGO_TRACE=1): s_memtime stamps of every wave of the last launch,
relative to the workgroup's own first entry stamp, in microseconds (shader clocks at ~2.4 GHz; TICK_US overrides).
  python tools/wgo_trace.py          (on the GPU box)
Rows: stamp name, then min / median / max over the waves that wrote it -- all workgroups, then the workgroups that carry a
small-parameter share (b < nsmall) and the others separately."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
os.environ["ACEZ_WGO_TRACE"] = "1"
from acezero_amd import _native as N

with N.diag_library():
    from tests.helpers import big_problem as _big_problem
    from tests.test_head_gpu import _trainer
    from tests import helpers
    from oracle import head_oracle
    prob = _big_problem()
    flat0 = head_oracle.init_params(3)
    cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
    cfg["global_batch"] = 5120
    cfg["iterations"] = 1000
    tr = _trainer(prob, flat0, cfg, max_batch=5120)
    rng = np.random.default_rng(1)
    batches = [torch.from_numpy(rng.permutation(prob["features"].shape[0])[:5120].astype(np.int64)).cuda() for _ in range(40)]
    for i in range(39):
        tr.step(batches[i], batches[i + 1])
    torch.cuda.synchronize()
    raw = np.zeros(256 * 12 * 8, np.uint64)
    N.check(tr.lib.acez_trainer_debug_read(tr._h, 7, 0, raw.ctypes.data_as(C.c_void_p), raw.nbytes, None))
tick_us = float(os.environ.get("TICK_US", str(1 / 2400.0)))   # s_memtime counts shader clocks (~2.4 GHz); the counters of different XCDs
t = raw.reshape(256, 12, 8).astype(np.int64)                  # have different origins, so every workgroup is timed from ITS first entry stamp
np.save(os.environ.get("WGO_TRACE_OUT", "/tmp/wgo_trace.npy"), t)
t0 = np.where(t[:, :, 0] > 0, t[:, :, 0], np.iinfo(np.int64).max).min(axis=1)[:, None, None]
rel = np.where(t > 0, (t - t0) * tick_us, np.nan)
names_m = ["entry", "K loop done", "sent / staged", "small params done", "(schedule wave) done"]
names_l = ["entry", "K loop done", "partner seen", "own half staged", "arithmetic done", "transpose rows in", "stores acknowledged"]
nsmall = 193
def show(title, sel, waves, names):
    print(title)
    for i, n in enumerate(names):
        v = rel[sel][:, waves, i].ravel()
        v = v[~np.isnan(v)]
        if v.size:
            print(f"  {n:24s} min {v.min():7.2f}  median {np.median(v):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}   ({v.size} waves)")
allb = np.arange(256)
show("multiplier waves, workgroups with a small-parameter share", allb[allb < nsmall], slice(0, 4), names_m)
show("multiplier waves, the other workgroups", allb[allb >= nsmall], slice(0, 4), names_m)
show("loader waves, workgroups with a small-parameter share", allb[allb < nsmall], slice(4, 12), names_l)
show("loader waves, the other workgroups", allb[allb >= nsmall], slice(4, 12), names_l)
print("schedule wave (workgroup 255, wave 0):", rel[255, 0, :5])

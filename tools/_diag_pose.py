import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from tests.test_head_gpu import _trainer
from tests.test_chain_gpu import _big_problem
from oracle import head_oracle

prob = _big_problem()
flat0 = head_oracle.init_params(helpers.SEED + 1)
cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_posemlp"], prob)
n = 2048
cfg["global_batch"] = n
def mk(**env):
    os.environ.update(env)
    try:
        return _trainer(prob, flat0, cfg, max_batch=5120)
    finally:
        for k in env: os.environ.pop(k)
trs = {"new": mk(ACEZ_CHAIN="0"), "old": mk(ACEZ_CHAIN="0", ACEZ_POSE_FUSED="0"), "chain": mk(ACEZ_CHAIN="1"), "chain_old": mk(ACEZ_CHAIN="1", ACEZ_POSE_FUSED="0"),
       "new2": mk(ACEZ_CHAIN="0")}
rng = np.random.default_rng(5)
for it in range(3):
    idx = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda()
    ref = trs["new"]
    for k, t in trs.items():
        if t is not ref:
            t.params.copy_(ref.params); t.adam_m.copy_(ref.adam_m); t.adam_v.copy_(ref.adam_v); t.sync_weights()
            t.pose_params.copy_(ref.pose_params); t.pose_m.copy_(ref.pose_m); t.pose_v.copy_(ref.pose_v)
    poses = {k: t.current_poses() for k, t in trs.items()}
    for t in trs.values():
        t.backward(idx)
    torch.cuda.synchronize()
    npar = ref.n_params
    g = {k: t.grad.cpu().numpy()[npar + 4:] for k, t in trs.items()}
    for k in trs:
        d = np.abs(g[k] - g["new"])
        print(it, k, "poses equal", np.array_equal(poses[k], poses["new"]), "pose grad equal", np.array_equal(g[k], g["new"]), "max abs diff", d.max(), "n diff", int((d > 0).sum()),
              "head grad equal", np.array_equal(t.grad.cpu().numpy()[:npar], ref.grad.cpu().numpy()[:npar]))
    for t in trs.values():
        t.update()
    torch.cuda.synchronize()

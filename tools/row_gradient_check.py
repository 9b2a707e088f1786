"""Per-row comparison of the gradient the head propagates (dZ of the last wide layer, debug read) with the CPU oracle, step by step, for one
golden configuration (default head_tanh_posemlp): which rows of the batch differ by more than 5 % and what share of the norm they carry.
Told a one-row event (an ill-conditioned patch whose gradient moves with the last bits of its refined pose) from a systematic error
when the 8-wave pose forward changed the summation order.   python tools/row_gradient_check.py [config]   (on the GPU box)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from tests import helpers
from tests.test_head_gpu import _trainer, _rel
from oracle import head_oracle
name = sys.argv[1] if len(sys.argv) > 1 else "head_tanh_posemlp"
prob, flat0, cfg = helpers.problem_for(name)
tr = _trainer(prob, flat0, cfg)
print("lib", tr.lib._name, "images", prob["image_pose_inv"].shape)
pose_flat = tr.pose_params.cpu().clone()
orc = head_oracle.TrainerOracle(flat0.clone(), prob["mean"], cfg, mode="bf16", pose_flat=pose_flat, image_pose_inv=prob["image_pose_inv"])
batches = helpers.golden_batches(prob, cfg["steps"])
n_params = flat0.numel()
for it, idx in enumerate(batches):
    b = helpers.torch_batch(prob, idx)
    di = torch.from_numpy(idx.astype(np.int64)).cuda()
    orc.head.p.flat.copy_(tr.params.cpu())
    orc.sched.m.copy_(tr.adam_m.cpu()); orc.sched.v.copy_(tr.adam_v.cpu())
    with torch.no_grad():
        orc.pose.flat.copy_(tr.pose_params.cpu())
    orc.pose_m.copy_(tr.pose_m.cpu()); orc.pose_v.copy_(tr.pose_v.cpu())
    dp = np.abs(tr.current_poses() - orc.current_poses().numpy()).max()
    cap = {}
    _bw = orc.head.backward
    def bw(tape, ds, _bw=_bw, cap=cap):
        f1i = 3 * (orc.head.nb + 1)
        cap["dz"] = orc.head.rg((ds @ orc.head.r(orc.head.p.W3)) * (tape["out"][f1i + 1] > 0)).numpy()
        cap["ds"] = ds.numpy().copy()
        return _bw(tape, ds)
    orc.head.backward = bw
    rec = orc.step(b["features"], b)
    orc.head.backward = _bw
    tr.backward(di)
    torch.cuda.synchronize()
    if rec is None:
        break
    grad = tr.grad.cpu().numpy()
    go = rec["grad"].numpy()
    print(it, "pose maxdiff %.2e" % dp, "head grad rel %.3e" % _rel(grad[:n_params], go), "pose grad rel %.3e" % _rel(grad[n_params + 4:], rec["pose_grad"].numpy()))
    n = len(idx)
    dzg = torch.from_numpy(tr.debug_read("dZ", tr.L - 1, n).astype(np.int32) << 16).view(torch.float32).numpy() / float(tr.state().get("grad_scale", 1.0) or 1.0)
    dzo = cap["dz"]
    rown = np.linalg.norm(dzg - dzo, axis=1); ref = np.linalg.norm(dzo, axis=1) + 1e-30
    bad = np.where(rown > 0.05 * np.maximum(ref, np.median(ref)))[0]
    print("   rows off by > 5%:", bad[:10], "of", n, "| their share of ||dz||:", float(np.linalg.norm(dzo[bad]) / np.linalg.norm(dzo)) if len(bad) else 0.0,
          "| scale check:", float(np.linalg.norm(dzg) / np.linalg.norm(dzo)))
    tr.update()

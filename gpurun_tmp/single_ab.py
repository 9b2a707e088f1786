import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.test_dp_gpu import _problem, _make_trainer, B
prob, flat0 = _problem()
n = prob["features"].shape[0]
gen = torch.Generator(device="cuda").manual_seed(8191)
perm = torch.randperm(n, generator=gen, device="cuda")
trs = []
for seq in ("0", "1"):
    os.environ["ACEZ_SEQ"] = seq
    trs.append(_make_trainer(prob, flat0, 0, n))
for b in range(3):
    gs = []
    for tr in trs:
        tr.backward(perm[b * B:(b + 1) * B].contiguous())
        gs.append(tr.grad.cpu().numpy().copy())
        tr.update()
    torch.cuda.synchronize()
    print("batch", b, "differing", int((gs[0] != gs[1]).sum()), "rel", float(np.linalg.norm(gs[0] - gs[1]) / np.linalg.norm(gs[0])), "params equal", torch.equal(trs[0].params, trs[1].params), flush=True)

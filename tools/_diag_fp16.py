import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from tests.test_head_gpu import _trainer
from tests.test_chain_gpu import _big_problem
from oracle import head_oracle
mode = sys.argv[1]
prob = _big_problem(n_images=8, patches_per_view=512)
flat0 = head_oracle.init_params(helpers.SEED + 1)
cfg = helpers.full_cfg(helpers.HEAD_CONFIGS["head_tanh_1cyclepoly"], prob)
cfg["global_batch"] = 2048
if mode == "fp16":
    trs = {}
    for name, seq in (("layers", "0"), ("chains", "1"), ("split", "1")):
        os.environ["ACEZ_SEQ"] = seq
        trs[name] = _trainer(prob, flat0, cfg, max_batch=2048, dtype="fp16")
        os.environ.pop("ACEZ_SEQ")
    rng = np.random.default_rng(3)
    for it in range(6):
        n = 2048 if it % 2 == 0 else 777
        idx = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda()
        for k, t in trs.items():
            if k == "split" or len(sys.argv) < 3:
                t.backward(idx)
            else:
                t.step(idx)
        torch.cuda.synchronize()
        ref = trs["layers"]
        for k, t in trs.items():
            eq_g = torch.equal(t.grad[:t.n_params], ref.grad[:t.n_params])
            L = ref.L
            bad = [l for l in range(L) if not np.array_equal(t.debug_read("dZ", l, n), ref.debug_read("dZ", l, n))]
            bado = [l for l in range(L) if not np.array_equal(t.debug_read("out", l, n), ref.debug_read("out", l, n))]
            print(it, k, "grad equal", eq_g, "dZ differ", bad, "out differ", bado, "stat", t.grad[t.n_params:t.n_params + 4].tolist(), "finite", bool(torch.isfinite(t.grad).all()))
        for k, t in trs.items():
            if k == "split" or len(sys.argv) < 3:
                t.update()
        torch.cuda.synchronize()
        for k, t in trs.items():
            print(it, k, "params equal", torch.equal(t.params, ref.params), t.state()["iteration"], t.state()["loss"])
else:
    # bf16: this build vs the round-2 library, bit for bit? (two processes: dump params after 10 fused steps)
    tr = _trainer(prob, flat0, cfg, max_batch=2048)
    rng = np.random.default_rng(3)
    idx = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:2048].astype(np.int64)).cuda()
    tr.backward(idx)
    torch.cuda.synchronize()
    g = tr.grad.cpu().numpy().copy()
    tr.update()
    torch.cuda.synchronize()
    p = np.concatenate([g, tr.params.cpu().numpy(), tr.last_scene_coords(2048).ravel()])
    out = sys.argv[2]
    np.save(out, p)
    print("saved", out, float(np.abs(p).sum()), tr.state())

import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import torch.multiprocessing as mp
from tests.test_dp_gpu import _dp_worker, _free_port

def run(seq):
    os.environ["ACEZ_SEQ"] = seq
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=200) for _ in procs], key=lambda t: t[0])
    for p in procs: p.join(timeout=60)
    return res

if __name__ == "__main__":
    a = run("0"); b = run("1"); c = run("1"); d = run("0")
    for name, x, y in (("seq0 vs seq1", a, b), ("seq1 vs seq1", b, c), ("seq0 vs seq0", a, d)):
        for k in range(3):
            g0, g1 = x[0][1][k], y[0][1][k]
            nd = int((g0 != g1).sum())
            print(name, "batch", k, "differing grad entries", nd, "rel", float(np.linalg.norm(g0 - g1) / np.linalg.norm(g0)), flush=True)
        print(name, "params equal", np.array_equal(x[0][2], y[0][2]), flush=True)

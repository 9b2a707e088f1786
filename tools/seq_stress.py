"""Two-process stress of rowseq_kernel (run on the GPU box):  python tools/seq_stress.py WORLD STEPS SEQ
WORLD processes share ONE GPU and step in lock step (a gloo barrier before every backward); each holds two trainers on the same
batches -- `tr` with ACEZ_SEQ=SEQ (and the XCD placement record) and `ref` with per-layer launches -- and counts the steps whose
gradient vectors differ in any bit; for the first such step it lists the buffers (layer, rows, columns) that differ. This is what
found the fragment-read race of DESIGN.md section 3 "Round 2" (0): seq vs per-layer 13 and 90 of 90 steps before the fix, 0 of
2 x 150 after; per-layer vs per-layer 0. Prints (rank, steps with siblings on different XCDs, differing steps, first difference)."""
import os as _os
_os.environ.setdefault("ACEZ_LIB", "diag")   # the ACEZ_* ablation switches exist in the diagnostics build only (acezero_amd/build.py --diag)
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp
from tests.test_dp_gpu import _problem, _make_trainer, _free_port, B

def xcc(tr):
    from acezero_amd import _native as N
    out = np.zeros(8 + 256, np.uint32)
    N.check(tr.lib.acez_trainer_debug_read(tr._h, 6, 0, out.ctypes.data_as(C.c_void_p), out.nbytes, None))
    return out

def worker(rank, world, port, q, steps, seqmode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    prob, flat0 = _problem()
    n = prob["features"].shape[0]
    os.environ["ACEZ_SEQ"] = seqmode; os.environ["ACEZ_SEQ_XCC"] = "1"
    tr = _make_trainer(prob, flat0, 0, n)
    os.environ["ACEZ_SEQ"] = "0"
    ref = _make_trainer(prob, flat0, 0, n)
    rng = np.random.default_rng(100 + rank)
    bad_place = 0; bad_grad = 0; first = None
    for it in range(steps):
        m = int(rng.integers(2000, 3000)) if it % 2 else 5120
        idx = torch.from_numpy(rng.permutation(n)[:m].astype(np.int64)).cuda()
        dist.barrier()                         # lock step: both ranks launch their chains at the same moment
        tr.backward(idx)
        torch.cuda.synchronize()
        if seqmode != "0":
            x = xcc(tr)
            sib = x[8:8 + 4 * ((m + 79) // 80)].reshape(-1, 4)
            bad_place += int(not (sib == sib[:, :1]).all())
        ref.backward(idx)
        torch.cuda.synchronize()
        if not torch.equal(tr.grad, ref.grad):
            bad_grad += 1
            if first is None:
                rep = []
                for kind, cnt in (("out", tr.L), ("R", tr.nb + 2), ("dZ", tr.L)):
                    for l in range(cnt):
                        a, b = tr.debug_read(kind, l, m), ref.debug_read(kind, l, m)
                        d = a != b
                        if d.any():
                            rows = np.nonzero(d.any(1))[0]; cols = np.nonzero(d.any(0))[0]
                            rep.append((kind, l, int(d.sum()), "rows %d..%d (tiles %s)" % (rows[0], rows[-1], sorted(set((rows // 80).tolist()))[:6]),
                                        "cols %d..%d" % (cols[0], cols[-1])))
                g1, g2 = tr.grad.cpu().numpy(), ref.grad.cpu().numpy()
                first = (it, m, float(np.linalg.norm(g1 - g2) / np.linalg.norm(g2)), rep[:6])
        tr.update(); ref.update()
    q.put((rank, bad_place, bad_grad, first))
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    world = int(sys.argv[1]); steps = int(sys.argv[2]); seqmode = sys.argv[3]
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q, steps, seqmode)) for r in range(world)]
    for p in procs: p.start()
    for _ in procs: print(q.get(timeout=300), flush=True)
    for p in procs: p.join(timeout=60)

"""Two-process stress of wgrad_opt_kernel's exchange (run on the GPU box):  python tools/wgo_stress.py WORLD STEPS
WORLD processes share ONE GPU and step in lock step (a gloo barrier before every step); each holds two trainers on the same batches --
`tr` (the default flow: wgrad_opt_kernel, next batch gathered beside the loss kernel) and `ref` (ACEZ_WGRAD_OPT=0: wgrad_kernel + the
optimiser launch) -- and takes fused steps with announced next batches. With a second tenant on the GPU the workgroups of a launch are
not all resident at once, which is exactly the situation the bounded poll, the fault word and the fall-back exist for. Prints per rank:
faults taken (seq_status), whether the two trainers' parameters and optimiser moments are bitwise equal (they must be whenever no fault
was taken; after a fault the faulting step is applied partially), the final losses, and NaN checks."""
import os as _os
_os.environ.setdefault("ACEZ_LIB", "diag")   # ACEZ_WGRAD_OPT exists in the diagnostics build only
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp
from tests.test_dp_gpu import _problem, _make_trainer, _free_port


def worker(rank, world, port, q, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    prob, flat0 = _problem()
    n = prob["features"].shape[0]
    os.environ.pop("ACEZ_WGRAD_OPT", None)
    tr = _make_trainer(prob, flat0, 0, n)
    os.environ["ACEZ_WGRAD_OPT"] = "0"
    ref = _make_trainer(prob, flat0, 0, n)
    os.environ.pop("ACEZ_WGRAD_OPT", None)
    rng = np.random.default_rng(100 + rank)
    batches = []
    for it in range(steps + 1):
        m = int(rng.integers(2000, 3000)) if it % 3 == 1 else 5120
        batches.append(torch.from_numpy(rng.permutation(n)[:m].astype(np.int64)).cuda())
    for it in range(steps):
        dist.barrier()                         # lock step: both ranks launch at the same moment
        tr.step(batches[it], batches[it + 1])
        ref.step(batches[it], batches[it + 1])
    torch.cuda.synchronize()
    st, sr = tr.state(), ref.state()
    same = bool(torch.equal(tr.params, ref.params) and torch.equal(tr.adam_m, ref.adam_m) and torch.equal(tr.adam_v, ref.adam_v))
    q.put((rank, {"faults": tr.seq_status()["faults"], "ref_faults": ref.seq_status()["faults"], "bitwise_equal": same,
                  "iteration": (st["iteration"], sr["iteration"]), "loss": (st["loss"], sr["loss"]),
                  "finite": bool(torch.isfinite(tr.params).all())}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]); steps = int(sys.argv[2])
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q, steps)) for r in range(world)]
    for p in procs: p.start()
    for _ in procs: print(q.get(timeout=300), flush=True)
    for p in procs: p.join(timeout=60)

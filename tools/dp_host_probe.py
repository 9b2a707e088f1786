"""Where a data-parallel step's time goes on ONE GPU with a one-rank RCCL group (no wire time): per mode, ms per step with the GPU drained
at the end (total) and ms per step until the host has ISSUED everything (host). host ~ total = the step is bound by the host's issue rate
(ctypes launches + torch ops + torch.distributed calls), not by the GPU.   python tools/dp_host_probe.py [steps]"""
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import bench
from acezero_amd import parallel, synth
from acezero_amd.head import HeadTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
n = 1_000_000
prob, feats, target_px, view_idx = bench.make_buffer(n, device, 2089)
perm = torch.randperm(n, generator=torch.Generator(device=device).manual_seed(8191), device=device)
batches = [perm[i * 5120:(i + 1) * 5120].contiguous() for i in range(n // 5120)]


def trainer():
    tr = HeadTrainer(prob["mean"], max_batch=5120, global_batch=5120, loss_type="tanh", schedule="1cyclepoly", iterations=25000, lr_min=0.0005,
                     lr_max=0.003, warmup_iterations=1000, warmup_lr=0.0005, cooldown_iterations=5000, dtype="bf16")
    tr.load_flat(torch.from_numpy(synth.init_head_params(1)))
    tr.set_buffer(feats, target_px, view_idx, prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"], prob["view_image"], prob["image_pose_inv"])
    return tr


def run(name, make_step, announce=False):
    tr = trainer()
    step = make_step(tr)
    if not announce:
        step1 = step
        step = lambda b, nxt: step1(b)
    for i in range(30):
        step(batches[i % len(batches)], batches[(i + 1) % len(batches)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(batches[(30 + i) % len(batches)], batches[(31 + i) % len(batches)])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-58s total %.1f us/step   host issue %.1f us/step" % (name, (t2 - t0) / steps * 1e6, (t1 - t0) / steps * 1e6), flush=True)
    tr.close()


def fused(tr):
    it = iter(range(10 ** 9))
    return lambda b: tr.step(b)


def split(tr):
    def s(b):
        tr.backward(b)
        tr.update()
    return s


def allreduce_rccl(tr):
    def s(b):
        tr.backward(b)
        dist.all_reduce(tr.grad)
        tr.update()
    return s


def sharded(tr, **kw):
    dp = parallel.ShardedDataParallel(tr, **kw)
    return dp.step


run("fused step (acez_train_step), no exchange", fused)
run("fused step, next batch announced (acez_train_step_next)", lambda tr: (lambda b, nxt: tr.step(b, nxt)), announce=True)
run("backward + update (split flow), no exchange", split)
run("split flow + ONE RCCL all_reduce of the 8.7 MB bucket", allreduce_rccl)
def split_announced(tr):
    def s(b, nxt):
        tr.backward(b)
        tr.update(nxt)
    return s


run("split flow, next batch announced to the update", split_announced, announce=True)
run("DEFAULT: DataParallelTrainer, one-rank RCCL all_reduce, next batch announced", lambda tr: parallel.DataParallelTrainer(tr, force_exchange=True).step, announce=True)
run("sharded flow, collectives skipped (proxy_world=1)", lambda tr: sharded(tr, proxy_world=1))
run("sharded flow, 3 RCCL collectives (force_exchange)", lambda tr: sharded(tr, force_exchange=True))
run("sharded flow, 3 RCCL collectives, all synchronous calls", lambda tr: sharded(tr, force_exchange=True, async_collectives=False))


def allreduce_async(tr):
    def s(b):
        tr.backward(b)
        dist.all_reduce(tr.grad, async_op=True).wait()
        tr.update()
    return s


def rs_only(tr):
    out = tr.grad.new_empty(8 * int(tr.LAYER_STRIDE))
    def s(b):
        tr.backward(b)
        dist.reduce_scatter_tensor(out, tr.grad[:8 * int(tr.LAYER_STRIDE)])
        tr.update()
    return s


run("split flow + ONE RCCL all_reduce, async_op + wait", allreduce_async)
run("split flow + ONE synchronous reduce_scatter_tensor (copy kernel)", rs_only)
run("sharded per-owner flow (reduce + broadcast per owner)", lambda tr: sharded(tr, force_exchange=True, one_shot=False))
dist.destroy_process_group()

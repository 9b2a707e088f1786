"""python tools/ipc_probe.py -- can two processes on this box share device memory through HIP IPC handles (hipIpcGetMemHandle /
hipIpcOpenMemHandle), for plain and for fine-grained allocations, and does a flag written by a kernel of one process become visible to a
kernel-side / host-side read of the other? (Feasibility probe for a peer-to-peer gradient exchange)"""
import ctypes as C
import os
import sys
import time

import multiprocessing as mp

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
def _find():
    # the SAME runtime torch (and therefore libacez.so) uses, if torch ships one; two HIP runtimes in a process do not share state
    import importlib.util
    sp = importlib.util.find_spec("torch")
    if sp and sp.submodule_search_locations:
        cand = os.path.join(list(sp.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            return cand
    return "libamdhip64.so"


hip = C.CDLL(_find())
HANDLE = C.c_char * 64


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hip error %d" % (what, rc))


def worker(rank, q01, q10, fine):
    check(hip.hipSetDevice(0), "hipSetDevice")
    if rank == 0:
        p = C.c_void_p()
        if fine:
            check(hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(1 << 20), C.c_uint(1)), "hipExtMallocWithFlags(finegrained)")   # hipDeviceMallocFinegrained = 1
        else:
            check(hip.hipMalloc(C.byref(p), C.c_size_t(1 << 20)), "hipMalloc")
        check(hip.hipMemset(p, 0, C.c_size_t(1 << 20)), "hipMemset")
        check(hip.hipDeviceSynchronize(), "sync")
        h = HANDLE()
        check(hip.hipIpcGetMemHandle(C.byref(h), p), "hipIpcGetMemHandle")
        q01.put(bytes(h.raw))
        assert q10.get(timeout=60) == "written"
        host = (C.c_uint32 * 4)()
        check(hip.hipMemcpy(host, p, 16, 2), "D2H")
        print("fine-grained" if fine else "plain", "allocation: owner reads", list(host), flush=True)
        q01.put("done")
    else:
        raw = q01.get(timeout=60)
        h = HANDLE.from_buffer_copy(raw)
        p = C.c_void_p()
        check(hip.hipIpcOpenMemHandle(C.byref(p), h, C.c_uint(1)), "hipIpcOpenMemHandle")   # hipIpcMemLazyEnablePeerAccess = 1
        vals = (C.c_uint32 * 4)(11, 22, 33, 44)
        check(hip.hipMemcpy(p, vals, 16, 1), "H2D into the peer's buffer")
        check(hip.hipDeviceSynchronize(), "sync")
        q10.put("written")
        assert q01.get(timeout=60) == "done"
        check(hip.hipIpcCloseMemHandle(p), "hipIpcCloseMemHandle")


if __name__ == "__main__":
    mp.set_start_method("spawn")
    for fine in (False, True):
        q01, q10 = mp.Queue(), mp.Queue()
        ps = [mp.Process(target=worker, args=(r, q01, q10, fine)) for r in range(2)]
        t0 = time.time()
        for p in ps:
            p.start()
        for p in ps:
            p.join(120)
        print(_find(), flush=True)
        print("fine" if fine else "plain", "exit codes", [p.exitcode for p in ps], "%.1f s" % (time.time() - t0), flush=True)

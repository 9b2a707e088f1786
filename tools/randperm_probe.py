import time, torch
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
for n in (614400, 8000000):
    for name, fn in (("randperm", lambda: torch.randperm(n, generator=g, device=dev)),
                     ("rand.argsort", lambda: torch.rand(n, generator=g, device=dev).argsort()),
                     ("randint32.sort", lambda: torch.sort(torch.randint(0, 2**31-1, (n,), generator=g, device=dev, dtype=torch.int32))[1])):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(5): p = fn()
        e1.record(); th = (time.perf_counter() - t0) / 5
        torch.cuda.synchronize()
        print(n, name, "gpu ms %.3f" % (e0.elapsed_time(e1) / 5), "host ms %.3f" % (th * 1e3), p.dtype)

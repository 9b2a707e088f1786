"""Timeline of convgemm512_kernel's tiles (diagnostics build): s_memtime stamps of the first multiplier wave and the first loader wave of every
256 x 256 tile of the LAST pointwise layer of a whole-frame head pass (acez_head_forward_maps on 64 frames), relative to the tile's own entry,
in microseconds (shader clocks at ~2.4 GHz; TICK_US overrides).   python tools/conv_trace.py   (on the GPU box)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from acezero_amd import synth, _native as N

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with N.diag_library() as lib:
    from acezero_amd.head import HeadTrainer
    n = F * 4800
    tr = HeadTrainer(np.zeros(3, np.float32), max_batch=n, iterations=1, inference_only=True)
    tr.load_flat(torch.from_numpy(synth.init_head_params(7)))
    f = (torch.randn(n, 512, device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty((F, 3, 60, 80), dtype=torch.float32, device="cuda")
    tiles = ((n + 255) // 256) * 2
    buf = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
    call = lambda: N.check(tr.lib.acez_head_forward_maps(tr._h, C.c_void_p(f.data_ptr()), F, 60, 80, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    lib.diagz_conv_trace.argtypes = [C.c_void_p]
    lib.diagz_conv_trace(C.c_void_p(buf.data_ptr()))
    call()
    torch.cuda.synchronize()
    lib.diagz_conv_trace(None)
    t = buf.cpu().numpy().reshape(tiles, 8)
tick = float(os.environ.get("TICK_US", str(1 / 2400.0)))
ok = t[:, 0] > 0
print(int(ok.sum()), "tiles of", tiles)
names = ["multiplier: entry", "multiplier: K loop done", "multiplier: epilogue tile complete", "multiplier: stores acknowledged",
         "loader: entry", "loader: first stage landed", "loader: K loop done (ring free)", "loader: epilogue tile complete"]
for i in (5, 1, 6, 2, 7, 3):
    v = (t[ok, i] - t[ok, 0]) * tick
    print(f"  {names[i]:36s} median {np.median(v):7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")

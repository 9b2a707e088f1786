import os, sys, json, argparse, torch
sys.path.insert(0, ".")
import bench
for ch in (32, 64, 128, 64, 32):
    os.environ["ACEZ_E2E_CHUNK"] = str(ch)
    args = argparse.Namespace(e2e_frames=1024)
    r = bench.bench_pipeline(args, 0, 1, torch.device("cuda", 0))
    print(ch, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}, "img/s", round(r["frames"] / r["e2e_s"]), "enc ms/frame", round(r["encoder_ms"] / r["frames"], 4), flush=True)

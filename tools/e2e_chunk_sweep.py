"""Frames per encoder pass of the bench's images -> poses leg: encoder ms per frame and end-to-end images/s for several chunk sizes
(ACEZ_E2E_CHUNK).  python tools/e2e_chunk_sweep.py [chunks...]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, ".")
import bench

for ch in [int(x) for x in sys.argv[1:]] or [32, 64, 128, 256]:
    os.environ["ACEZ_E2E_CHUNK"] = str(ch)
    r = bench.bench_pipeline(argparse.Namespace(e2e_frames=max(1024, 4 * ch)), 0, 1, torch.device("cuda", 0))
    enc = r["encoder_ms"] / r["frames"]
    print(ch, "frames per pass: encoder", round(enc, 4), "ms per frame =", round(bench.ENC_FLOP_PER_FRAME / (enc * 1e-3) / 1e12 / bench.MFMA_PEAK_TFLOPS, 3),
          "of the MFMA peak; end to end", round(r["frames"] / r["e2e_s"]), "images/s; buffer", round(r["buffer_rows"] / r["buffer_s"] / 1e6, 1), "M rows/s", flush=True)

"""The bench's in-process reconstruction leg with the session's INFO log: python tools/session_timing.py [frames]"""
import logging
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

logging.basicConfig(level=logging.INFO)
torch.cuda.set_device(0)
t0 = time.perf_counter()
out = bench.bench_session(SimpleNamespace(session_frames=int(sys.argv[1]) if len(sys.argv) > 1 else 120), torch.device("cuda", 0))
print({k: v for k, v in out.items() if k not in ("note", "metric")}, "wall incl. render %.2f s" % (time.perf_counter() - t0))

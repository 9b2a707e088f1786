import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import chain_timing
for c in ("0", "0", "1", "0"):
    chain_timing.run(c, 1_000_000, 300)

"""A/B of the two wgrad tilings (default: 128x128 workgroup tiles, 2 slabs; ACEZ_WGRAD_TILE=256: 256x128 tiles of four 128x64 waves, 4 slabs), same process and buffer."""
import os as _os
_os.environ.setdefault("ACEZ_LIB", "diag")   # the ACEZ_* ablation switches exist in the diagnostics build only (acezero_amd/build.py --diag)
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import chain_timing  # noqa: E402

if __name__ == "__main__":
    for tile in ("128", "256", "128", "256"):
        os.environ["ACEZ_WGRAD_TILE"] = tile
        print("ACEZ_WGRAD_TILE=" + tile, end="  ")
        chain_timing.run("0", 1_000_000, 300)

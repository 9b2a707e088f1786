import os as _os
_os.environ.setdefault("ACEZ_LIB", "diag")   # the ACEZ_* ablation switches exist in the diagnostics build only (acezero_amd/build.py --diag)
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import chain_timing
for d in ("0", "1", "2"):
    os.environ["ACEZ_LOSS_DBG"] = d
    print("ACEZ_LOSS_DBG=" + d, end="  ")
    chain_timing.run("0", 1000000, 200)

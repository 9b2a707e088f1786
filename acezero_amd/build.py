"""Build libacez.so (HIP, gfx950) in-tree with hipcc.  `python -m acezero_amd.build [--force]`.

The library is the product: there is no CPU fallback.  hipcc cross-compiles for gfx950 without a GPU, so this
also runs in the CPU-only build container.  The built .so is git-ignored but travels to the GPU box.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libacez.so")
STAMP = os.path.join(HERE, ".libacez.stamp")
# the diagnostics build (-DACEZ_DIAG): ablation switches, measured-and-rejected kernels, fault-injection hooks. tests/ and tools/ only
# (acezero_amd._native.diag_library()); nothing in the product loads it.
LIB_DIAG = os.path.join(HERE, "libacez_diag.so")
STAMP_DIAG = os.path.join(HERE, ".libacez_diag.stamp")

# translation unit -> extra flags.  The RANSAC and point-cloud units must not contract a*b+c into fma: their arithmetic
# is compared bit-for-bit with the CPU oracle (DESIGN.md "Determinism").
UNITS = {
    "acez_common.hip": [],
    "head_api.hip": [],
    "encoder_api.hip": [],
    "ransac_api.hip": ["-ffp-contract=off"],
    "cloud_api.hip": ["-ffp-contract=off"],
}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _digest():
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for name in sorted(os.listdir(root)):
            p = os.path.join(root, name)
            if os.path.isfile(p):
                h.update(name.encode())
                with open(p, "rb") as f:
                    h.update(f.read())
    h.update(repr(sorted(UNITS.items())).encode())
    h.update(repr(COMMON).encode())
    return h.hexdigest()


def build(force=False, verbose=False, diag=False):
    """Compile every HIP unit for gfx950 and link libacez.so (diag=True: libacez_diag.so, the same sources with -DACEZ_DIAG).
    Returns the library path."""
    LIB, STAMP = (LIB_DIAG, STAMP_DIAG) if diag else (globals()["LIB"], globals()["STAMP"])
    digest = _digest() + ("-diag" if diag else "")
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == digest:
                return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build_diag" if diag else "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for unit, extra in UNITS.items():
        src = os.path.join(CSRC, unit)
        if not os.path.exists(src):
            raise RuntimeError(f"missing source {src}")
        obj = os.path.join(objdir, unit.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + COMMON + extra + (["-DACEZ_DIAG"] if diag else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout))
    with open(STAMP, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))

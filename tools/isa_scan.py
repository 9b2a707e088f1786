"""Static scan of the compiled kernels (no GPU needed): for every kernel with MFMA instructions, the innermost loop (or unrolled
span) that contains them and what else sits in it -- LDS reads, full lgkmcnt waits, scratch (register spill) accesses, global
loads, s_nop -- plus every kernel's private-segment size and spill counts. Found the in-loop spills of conv3x3p_kernel
and the scratch-resident arrays of the RANSAC kernel.

    python tools/isa_scan.py            # compiles acezero_amd/csrc/*.hip with -save-temps into /tmp/acez_isa and scans the .s files
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acezero_amd import build as B  # noqa: E402

OUT = "/tmp/acez_isa"
P_LGKM0 = r"s_waitcnt.*lgkmcnt\(0\)"


def compile_units():
    os.makedirs(OUT, exist_ok=True)
    for unit, extra in B.UNITS.items():
        cmd = [B._hipcc()] + B.COMMON + extra + ["-save-temps=obj", "-c", os.path.join(B.CSRC, unit), "-o", os.path.join(OUT, unit.replace(".hip", ".o"))]
        subprocess.run(cmd, cwd=OUT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)


def scan(path):
    text = open(path).read()
    lines = text.split("\n")
    unit = os.path.basename(path).split("-hip-")[0]
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text):
        if int(m.group(2)) or int(m.group(4)):
            print("%-12s %-60s scratch %5s B/thread  vgpr %s  spilled %s" % (unit, m.group(1)[:60], m.group(2), m.group(3), m.group(4)))
    starts = [(i, re.match(r"^(_Z\S+):", l).group(1)) for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l)]
    for start, name in starts:
        end = next((i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i]), None)
        if end is None:
            continue
        body = lines[start:end]
        mf = [i for i, l in enumerate(body) if "v_mfma" in l]
        if not mf:
            continue
        labels = {re.match(r"^(\.LBB\d+_\d+):", l).group(1): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        loops = set()
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.add((labels[m.group(1)], i))
        ml = [(a, b) for a, b in loops if any(a <= j <= b for j in mf)]
        a, b = min(ml, key=lambda ab: ab[1] - ab[0]) if ml else (mf[0], mf[-1])
        seg = body[a:b + 1]
        cnt = lambda pat: sum(1 for l in seg if re.search(pat, l))  # noqa: E731
        print("%-12s %-48s %-8s span %5d mfma %4d ds_read %4d lgkmcnt(0) %3d scratch %2d global_load %3d s_nop %3d" % (
            unit, re.sub(r"^_ZN\d*[a-z_]*\d+", "", name)[:48], "loop" if ml else "unrolled", b - a, cnt("v_mfma"), cnt("ds_read"), cnt(P_LGKM0),
            cnt("scratch_"), cnt("global_load_dword"), cnt("s_nop")))


def scan_serial_loads(path):
    """Kernels whose global loads are each followed by a FULL wait: `global_load ...; s_waitcnt vmcnt(0)` pairs are serial memory
    round trips (~1 us each under load). Found loss_kernel's 12 + 32 of them in round 2 (19.7 -> 15.7 us once its loads were issued
    up front). A kernel is listed when it has >= 6 plain global loads and at least half as many full waits as loads."""
    unit = os.path.basename(path).split("-hip-")[0]
    name, loads, w0, wn = None, 0, 0, 0
    for ln in open(path):
        m = re.match(r"^(_Z\S+):", ln)
        if m:
            name, loads, w0, wn = m.group(1), 0, 0, 0
        if not name:
            continue
        if "global_load" in ln and "lds" not in ln:
            loads += 1
        if re.search(r"s_waitcnt.*vmcnt\(0\)", ln):
            w0 += 1
        elif re.search(r"s_waitcnt.*vmcnt\(\d+\)", ln):
            wn += 1
        if "s_endpgm" in ln:
            if loads >= 6 and 2 * w0 >= loads:
                print("%-12s %-60s global loads %3d  vmcnt(0) waits %3d  counted waits %3d   <- serial round trips?" % (
                    unit, re.sub(r"^_ZN\d*[a-z_]*\d+", "", name)[:60], loads, w0, wn))
            name = None


def scan_reads_across_barrier(path):
    """Kernels that refill LDS asynchronously (LDS-DMA) and pass an s_barrier with ds_reads still in flight: if that barrier is the
    one that releases the refill of the slot being read, only the DMA latency protects the reads. Found in rowseq_kernel in round 2
    (straight-line K loop: the scheduler had sunk the MFMAs and their waits below the barrier; a rounding-level corruption once in
    ~50 steps with two processes on one GPU). Linear walk, lgkmcnt(N) leaves at most N reads pending. chain_kernel is listed by design:
    it prefetches the next pair's W fragments into registers across the barrier and its ring refills a slot one pair later."""
    unit = os.path.basename(path).split("-hip-")[0]
    name, pending, hits, dma, barriers = None, 0, 0, 0, 0
    for ln in open(path):
        m = re.match(r"^(_Z\S+):", ln)
        if m:
            name, pending, hits, dma, barriers = m.group(1), 0, 0, 0, 0
        if not name:
            continue
        if re.search(r"\bds_read", ln):
            pending += 1
        elif "global_load_lds" in ln:
            dma += 1
        else:
            w = re.search(r"s_waitcnt.*lgkmcnt\((\d+)\)", ln)
            if w:
                pending = min(pending, int(w.group(1)))
            elif re.search(r"\bs_barrier\b", ln):
                barriers += 1
                hits += pending > 0
        if "s_endpgm" in ln:
            if dma and hits:
                print("%-12s %-60s barriers %3d, with ds_reads in flight %3d   <- is the refill of their slot released by it?" % (
                    unit, re.sub(r"^_ZN\d*[a-z_]*\d+", "", name)[:60], barriers, hits))
            name = None


if __name__ == "__main__":
    compile_units()
    for f in sorted(os.listdir(OUT)):
        if f.endswith("gfx950.s"):
            scan(os.path.join(OUT, f))
    for f in sorted(os.listdir(OUT)):
        if f.endswith("gfx950.s"):
            scan_serial_loads(os.path.join(OUT, f))
    for f in sorted(os.listdir(OUT)):
        if f.endswith("gfx950.s"):
            scan_reads_across_barrier(os.path.join(OUT, f))

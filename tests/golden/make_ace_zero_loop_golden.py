"""Capture the control flow of the REFERENCE's ace_zero.py main loop: which train_ace.py / register_mapping.py commands it
issues, with which flags, for scripted registration outcomes -> tests/golden/ace_zero_loop.json.

Run in the build container only (needs /root/reference):   python tests/golden/make_ace_zero_loop_golden.py

ace_zero.py runs unmodified through runpy; ace_zero_util.run_cmd (the subprocess launcher) is replaced by a recorder that, for a
register_mapping.py command, writes the poses_<session>.txt file the loop reads its registration rate from (confidences
scripted per scenario). dataset_io (ZoeDepth download, pose-file parsing for the refined focal length) is a stub that returns a
scripted focal length. tests/test_session_cpu.py replays the same scenarios through acezero_amd.session.ReconstructionSession
(map / register replaced by the same scripted outcomes) and requires the same sequence of decisions.
"""
import json
import os
import runpy
import sys
import tempfile
import types
from pathlib import Path

REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
N_IMAGES = 200

# scenario -> (extra argv, registration rates in the order the register commands are issued:
#              seed 0 fast check, seed 1 fast check, best seed on all images, iteration 1, iteration 2, ...)
SCENARIOS = {
    "reaches_threshold": ([], [0.05, 0.10, 0.10, 0.50, 0.995, 1.0, 1.0]),
    "relative_threshold": ([], [0.10, 0.05, 0.10, 0.30, 0.305, 0.31, 0.31]),
    "no_final_refine": (["--final_refine", "False"], [0.05, 0.10, 0.10, 0.50, 0.995, 1.0]),
    "no_final_refit": (["--final_refit", "False"], [0.05, 0.10, 0.10, 0.50, 0.995, 1.0, 1.0]),
    "iterations_max": (["--iterations_max", "4"], [0.05, 0.10, 0.10, 0.30, 0.50, 0.70, 0.90, 0.95]),
    "no_warmstart": (["--warmstart", "False"], [0.05, 0.10, 0.10, 0.40, 0.80, 0.995, 1.0, 1.0]),
    "naive_refinement_no_calibration": (["--refinement", "naive", "--refine_calibration", "False"], [0.2, 0.1, 0.2, 0.995, 1.0, 1.0]),
    "slow_growth": ([], [0.02, 0.01, 0.02, 0.10, 0.25, 0.45, 0.60, 0.605, 0.61, 0.61]),
    # a pre-trained network instead of seed trials: the first register command uses it, mapping round 1 warm-starts from it
    "seed_network": (["--seed_network", "SEEDDIR/seed_network.pt"], [0.30, 0.60, 0.995, 1.0, 1.0]),
}


def flags_of(cmd):
    out, i = {}, 3
    while i < len(cmd):
        out[cmd[i].lstrip("-")] = cmd[i + 1]
        i += 2
    return out


def run_scenario(name):
    import ace_zero_util as zutil
    extra, rates = SCENARIOS[name]
    calls, reg_count = [], [0]
    tmp = Path(tempfile.mkdtemp())

    def fake_run_cmd(cmd, raise_on_error=True, verbose=True):
        cmd = [str(c) for c in cmd]
        fl = flags_of(cmd)
        if cmd[0] == zutil.TRAINING_EXE:
            calls.append({"cmd": "train", "id": Path(cmd[2]).stem, "flags": fl})
        elif cmd[0] == zutil.REGISTER_EXE:
            rate = rates[reg_count[0]]
            reg_count[0] += 1
            k = round(rate * N_IMAGES)
            with open(Path(cmd[2]).parent / f"poses_{fl['session']}.txt", "w") as f:
                for i in range(N_IMAGES):
                    f.write(f"img{i}.png 1 0 0 0 0 0 0 525.0 {1000 if i < k else 0}\n")
            calls.append({"cmd": "register", "network": Path(cmd[2]).stem, "flags": fl})
        else:
            calls.append({"cmd": cmd[0], "flags": {}})
        return 0

    zutil.run_cmd = fake_run_cmd
    dio = types.ModuleType("dataset_io")
    dio.get_depth_model = lambda init=False: object()
    focal_calls = [0]

    def load_dataset_ace(pose_file, confidence_threshold):
        focal_calls[0] += 1
        return [], [], [500.0 + focal_calls[0]]       # the "refined focal length" of mapping round k is 500 + k
    dio.load_dataset_ace = load_dataset_ace
    sys.modules["dataset_io"] = dio
    argv = sys.argv
    sys.argv = ["ace_zero.py", "scene/*.png", str(tmp), "--seed_parallel_workers", "1", "--try_seeds", "2"] + [e.replace("SEEDDIR", str(tmp)) for e in extra]
    try:
        runpy.run_path(os.path.join(REF, "ace_zero.py"), run_name="__main__")
    finally:
        sys.argv = argv
    for c in calls:                                   # paths inside the temporary folder -> bare names
        c["flags"] = {k: (Path(v).name if str(tmp) in v else v) for k, v in c["flags"].items()}
    return {"argv": extra, "rates": rates, "registers_used": reg_count[0], "calls": calls}


if __name__ == "__main__":
    out = {name: run_scenario(name) for name in SCENARIOS}
    with open(os.path.join(HERE, "ace_zero_loop.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for name, r in out.items():
        print(name, r["registers_used"], [(c["cmd"], c.get("id") or c.get("network")) for c in r["calls"]])

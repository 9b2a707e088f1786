"""CPU: the JSON line bench.py prints (the driver's contract) assembled from mocked measurements -- every field the contract and
the tier framing name is present and consistent; guards the assembly code, which otherwise only runs on the GPU box."""
import io
import json
import sys
from contextlib import redirect_stdout

import pytest
import torch


@pytest.mark.parametrize("seq", ["1", "0"])
def test_bench_json_line_contract(monkeypatch, seq):
    import bench
    prof = {k: [v, n] for k, (v, n) in {"sched": (0.0, 0), "gather": (0.14, 20), "gemm_fwd": (0.88, 160), "loss": (0.34, 20),
                                        "gemm_dgrad": (0.94, 140), "wgrad": (0.58, 20), "grad_reduce": (0.0, 0), "adamw": (0.42, 20)}.items()}
    monkeypatch.setenv("ACEZ_SEQ", seq)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    wins = [0.1551, 0.1549, 0.1550, 0.1620, 0.1548]             # five timed windows: the median is quoted, the mean gives `value`
    monkeypatch.setattr(bench, "bench_training", lambda *a, **k: (300 * sum(wins) / 5 * 1e-3 if not k.get("steps") else k["steps"] * 240e-6,
                                                                  {"loss": 22.3, "focal_scale": 1.01, "window_ms_per_step": list(wins)}, prof))
    monkeypatch.setattr(bench, "bench_registration", lambda *a, **k: (2048, 2048 / 340e3, 1.0))
    monkeypatch.setattr(bench, "bench_dp_rank_proxy", lambda args, device, rows, **k: 0.150 if rows == 5120 else 0.110)
    monkeypatch.setattr(bench, "dp_world1_legs", lambda args: ({"allreduce_rccl": 0.150, "sharded_skipped": 0.150, "sharded_rccl": 0.150,
                                                                 "sharded_rank_of_8": 0.150}, None))   # (a child process on the GPU box)
    monkeypatch.setattr(bench, "bench_pipeline", lambda *a, **k: {"frames": 256, "e2e_s": 0.025, "encoder_ms": 16.0, "buffer_rows": 262144, "buffer_s": 0.017,
                                                            "cloud_frames": 256, "cloud_s": 1.4e-4, "cloud_points": 256000})
    monkeypatch.setattr(bench, "bench_session", lambda *a: {"frames": 120, "seconds": 5.4})
    monkeypatch.setattr(bench, "cpu_baseline", lambda: {"value": 5.9e4, "unit": "patches/s", "cores": 32, "kind": "port", "sample": "mock"})
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    out = io.StringIO()
    with redirect_stdout(out):
        bench.main()
    lines = [ln for ln in out.getvalue().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                       # ONE JSON line
    d = json.loads(lines[0])
    assert d["metric"] == "ACE patches/sec" and d["unit"] == "patches/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["steps"] == 300 and d["warmup"] == 30 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert abs(d["value"] - 5120 / (sum(wins) / 5 * 1e-3)) < 1 and abs(d["ms_per_step"] - 0.1550) < 1e-9 and "workload" in d["config"]
    assert d["windows"] == 5 and d["window_ms_per_step"] == wins and abs(d["ms_per_step_mean"] - sum(wins) / 5) < 1e-9
    assert d["garden_like"]["training"]["n_images"] == 185 and d["garden_like"]["registration"]["map"] == "60x93"
    assert d["refinement_step"]["ms_per_step"] > 0 and "mfma_busy_frac" in d["roofline"] and d["dtype_fp16"]["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert ("rowseq_kernel" in r["kernel"]) == (seq == "1") and r["launches_timed"] == 300
    per_layer_us = (0.88 + 0.94) / 300 * 1e3
    assert abs(r["avg_launch_us"] - per_layer_us) < 1e-9 and abs(r["achieved"] - 2 * 5120 * 512 * 512 / (per_layer_us * 1e-6) / 1e12) < 1e-6
    # bytes per layer: a stored counter pass, quoted only while its source digest matches the running build (else null, never stale)
    assert r["traffic"] is None or (r["traffic"] > 1e6 and r["stored_profile"]["source_digest"] == bench.src_digest(bench.STEP_SOURCES))
    rr = d["roofline_ransac"]
    assert rr["frac"] is None or rr["stored_profile"]["source_digest"] == bench.src_digest(bench.RANSAC_SOURCES)
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["roofline_wgrad"]["frac"] > 0 and "roofline_ransac" in d
    # round 4: the tolerance table the parity tests assert, the longest single kernel beside the averaged figure, and the recipe to recompute
    assert d["parity"]["scene_coordinates_rel"] == {"bf16_vs_oracle_bf16": 1e-3, "bf16_vs_reference_fp32": 3e-2, "fp16_vs_reference_fp32": 2e-3}
    dom = r["dominant_kernel"]
    assert abs(dom["us_per_step"] - 0.94 / 20 * 1e3) < 1e-9 and dom["layers"] == 7
    assert abs(dom["frac"] - 7 * 2 * 5120 * 512 * 512 / (0.94 / 20 * 1e-3) / 1e12 / 2500.0) < 1e-9
    sp = r["stored_step_profile"]
    assert sp is None or (sp["algorithmic_bytes_per_step"] == bench.ALGO_BYTES_PER_STEP and sp["counter_bytes_per_step"] > 0)
    assert 60e6 < bench.ALGO_BYTES_PER_STEP < 70e6            # ~65 MB per step
    # round 5: the figures a reader needs first, as flat scalars inside `roofline` (the driver's record keeps those) and as `summary`
    sm = d["summary"]
    for k in ("dominant_kernel_frac", "refinement_ms_per_step", "fp16_ms_per_step", "registration_images_per_s", "registration_e2e_images_per_s",
              "encoder_frac_of_mfma_peak", "ransac_algorithmic_frac", "ransac_issue_occupancy"):
        assert k in sm and r["summary_" + k] == sm[k]
    assert abs(sm["dominant_kernel_frac"] - dom["frac"]) < 1e-12 and sm["registration_images_per_s"] == d["registration"]["value"]
    assert list(d).index("summary") < list(d).index("roofline")
    assert abs(rr["algorithmic_frac"] - rr["algorithmic"]["scoring_frac_of_fp64_valu_peak"]) < 1e-15 and (rr["frac"] is None or "occupancy" in rr["frac_is"])
    # round 6: the reference's precision end to end, the one-GPU proxy of a data-parallel rank, the rank count of the process group
    for k in ("refinement_fp16_ms_per_step", "e2e_fp16_images_per_s", "dp_rank_compute_ms_5120", "dp_rank_compute_ms_640"):
        assert k in sm and r["summary_" + k] == sm[k] and sm[k] > 0
    assert d["rccl_ranks"] == 0 and d["dp_rank_proxy"]["rows_640_ms"] == 0.110 and d["dtype_fp16"]["registration_e2e_images_per_s"] > 0
    # the RCCL calls themselves on a one-rank group (the only hardware run of that branch a one-GPU box allows)
    w1 = d["dp_rank_proxy"]["world1_exchange"]
    assert w1["allreduce_rccl_ms"] == 0.150 and w1["sharded_rccl_ms"] == 0.150 and sm["dp_world1_exchange_rccl_ms"] == r["summary_dp_world1_exchange_rccl_ms"] == 0.150
    assert d["dp_rank_proxy"]["mode"] == "allreduce"


def test_gpus_flag_without_a_launcher_re_executes_under_torchrun(monkeypatch):
    """`python bench.py --gpus 8` (the shape of the driver's N = 1 command) used to run ONE rank and print n_gpus 1. Now it
    re-executes itself under torch.distributed.run with 8 ranks -- or refuses when the node has fewer GPUs; a launcher whose WORLD_SIZE
    disagrees with --gpus is refused as well."""
    import subprocess
    import bench
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # fewer GPUs than ranks asked for: loud refusal, nothing launched
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    seen.clear()
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "only 1 GPU" in str(e.value.code) and not seen
    # a launcher's environment that disagrees with --gpus
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "must agree" in str(e.value.code)


def test_cpu_baseline_quotes_the_stored_reference_figure():
    """The reference's own training_step cannot run on the GPU box; its figure from the build container is stored with its core count
    (profiles/r04_cpu_reference_training_step.json) and quoted beside the live port, never instead of it."""
    import json as js
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rs = js.load(open(os.path.join(root, "profiles", "r04_cpu_reference_training_step.json")))
    assert rs["patches_per_s"] > 1e3 and rs["threads"] >= 1 and "training_step" in rs["what"]
    src = open(os.path.join(root, "bench.py")).read()
    assert "reference (stored)" in src and "r04_cpu_reference_training_step.json" in src


def test_stdout_carries_only_the_json_line_whatever_libraries_write_to_descriptor_1():
    """RCCL prints a version banner to file descriptor 1 at the first communicator of a process (seen on the GPU box: five lines behind
    the JSON line). bench.claim_stdout() keeps the real stdout for the line and sends descriptor 1 to stderr."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); import bench; out = bench.claim_stdout(); os.write(1, b'RCCL version : banner\\n'); "
            "print('python-level chatter'); print('{\"metric\": 1}', file=out, flush=True)" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"metric": 1}\n'
    assert "RCCL version : banner" in r.stderr and "python-level chatter" in r.stderr


def test_one_rank_rccl_legs_cannot_take_the_line_down(monkeypatch):
    """The legs that form a one-rank RCCL group run in a child process under a time limit; a failing or hanging child yields (None, error)."""
    import subprocess
    import types
    import bench
    args = types.SimpleNamespace(buffer_patches=1000)
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=1, stdout="", stderr="RCCL bootstrap failed"))
    out, err = bench.dp_world1_legs(args)
    assert out is None and "rc 1" in err and "bootstrap" in err

    def hang(*a, **k):
        raise subprocess.TimeoutExpired(a[0], k.get("timeout"))
    monkeypatch.setattr(subprocess, "run", hang)
    out, err = bench.dp_world1_legs(args)
    assert out is None and "TimeoutExpired" in err
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=0, stdout='RCCL version : x\n{"allreduce_rccl": 0.16}\n', stderr=""))
    assert bench.dp_world1_legs(args) == ({"allreduce_rccl": 0.16}, None)

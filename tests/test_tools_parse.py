"""CPU: the measurement tools under tools/ (run on the GPU box through gpurun, where a typo costs GPU minutes) at least parse:
Python files compile, shell recipes pass `bash -n`, stand-alone HIP programs pass `hipcc -fsyntax-only` for gfx950."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))), ids=os.path.basename)
def test_python_tools_compile(path):
    compile(open(path).read(), path, "exec")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))), ids=os.path.basename)
def test_shell_recipes_parse(path):
    assert subprocess.run(["bash", "-n", path], capture_output=True, text=True).returncode == 0


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tools", "*.hip"))), ids=os.path.basename)
def test_hip_tools_pass_the_front_end(path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "acezero_amd", "csrc"),
                        "-I", os.path.join(ROOT, "include"), path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]

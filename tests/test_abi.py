"""CPU: libacez.so builds with hipcc for gfx950, loads, and exports every symbol include/acez.h declares.
No compute is attempted: without a GPU the entry points must fail loudly (ACEZ_ERR_NODEVICE), never fall back."""
import ctypes as C
import os
import re

import pytest
import torch

from acezero_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "acez.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(acez_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(N.SYMBOLS), declared ^ set(N.SYMBOLS)
    lib = N.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.acez_version()


def test_library_contains_gfx950_code_objects():
    blob = open(os.path.join(ROOT, "acezero_amd", "libacez.so"), "rb").read()
    assert b"gfx950" in blob
    for kern in (b"rowgemm_kernel", b"wgrad_kernel", b"loss_kernel", b"adamw_kernel", b"ransac_kernel"):
        assert kern in blob, kern


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_silent_cpu_fallback():
    lib = N.lib()
    assert lib.acez_device_count() == 0
    h = C.c_void_p()
    assert lib.acez_ransac_create(C.byref(h), 1, 60, 80, -1) == -3
    assert b"no HIP device" in lib.acez_last_error()
    from acezero_amd import dsacstar, head
    with pytest.raises(RuntimeError):
        dsacstar.forward_rgb(torch.zeros(1, 3, 60, 80), torch.zeros(4, 4), 8, 10.0, 525.0, 320.0, 240.0, 100.0, 100.0, 8, 1, 16)
    with pytest.raises(RuntimeError):
        head.HeadTrainer([0, 0, 0])


def test_argument_validation_without_device():
    lib = N.lib()
    assert lib.acez_ransac_create(None, 1, 60, 80, -1) == -1
    hd = N.HeadDesc(1, 1, (C.c_float * 3)(0, 0, 0), 0.25, 100.0, 0.9)
    assert lib.acez_head_num_params(C.byref(hd)) == 2103300
    hd0 = N.HeadDesc(0, 0, (C.c_float * 3)(0, 0, 0), 0.25, 100.0, 0.9)
    assert lib.acez_head_num_params(C.byref(hd0)) == 5 * 262656 + 3 * 513

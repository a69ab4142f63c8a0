"""The C-ABI shared library loads without a GPU and exports every function include/ptrt.h declares;
device-needing calls fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "ptrt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ptrt_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(P):
    names = declared()
    assert len(names) >= 20 and "ptrt_render" in names and "ptrt_create" in names
    for n in names:
        assert hasattr(P.lib, n), f"{n} declared in include/ptrt.h but not exported"


def test_struct_layouts_match_reference_sizes(P):
    # vec3 12, DeviceBVHNode 40, Tri 12, Light 60, HitInfo 64 bytes
    assert ctypes.sizeof(P.Vec3) == 12 and ctypes.sizeof(P.BvhNode) == 40 and ctypes.sizeof(P.Tri) == 12
    assert ctypes.sizeof(P.Light) == 60 and ctypes.sizeof(P.Hit) == 64
    assert P.Light.position.offset == 4 and P.Light.color.offset == 28 and P.Light.radius.offset == 56
    assert P.lib.ptrt_abi_version() == 3


def test_no_cpu_fallback(P):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; this checks the GPU-less behaviour")
    ctx = ctypes.c_void_p()
    rc = P.lib.ptrt_create(64, 64, 0, 0, 0, ctypes.byref(ctx))
    assert rc == -2 and not ctx.value  # PTRT_E_NO_DEVICE
    assert b"no CPU path" in P.lib.ptrt_last_error(None)
    with pytest.raises(P.PtrtError):
        P.Scene(64, 64, device=0)
    s = P.Scene(64, 64, device=P.HOST_ONLY)
    P.scenes.cornell(s)
    with pytest.raises(P.PtrtError):
        s.uploadToGPU()
    P.lib.ptrt_destroy(None)  # tolerated


def test_bad_arguments_are_rejected(P):
    ctx = ctypes.c_void_p()
    assert P.lib.ptrt_create(0, 64, 0, 0, 0, ctypes.byref(ctx)) == -1
    assert P.lib.ptrt_create(64, 64, 60, 10, 0, ctypes.byref(ctx)) == -1
    assert P.lib.ptrt_render(None, 0, 1, 1, None, 0) == -1

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
    assert P.lib.ptrt_abi_version() == 6


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


def test_stale_and_foreign_handles_are_refused_without_touching_them(P):
    """A destroyed (or never created) handle must be answered with PTRT_E_INVALID and its memory left alone
    (fail() used to write the message into the freed context)."""
    buf = ctypes.create_string_buffer(4096)  # stands in for a freed ptrt_ctx: never in the live set
    stale = ctypes.cast(buf, ctypes.c_void_p)
    before = bytes(buf.raw)
    assert P.lib.ptrt_render(stale, 0, 1, 1, None, 0) == -1
    assert P.lib.ptrt_reset_rng(stale, 1) == -1
    assert P.lib.ptrt_refit(stale) == -1
    assert P.lib.ptrt_sync(stale) == -1
    assert P.lib.ptrt_set_option(stale, b"count_rays", 1) == -1
    assert bytes(buf.raw) == before, "an entry point wrote into a handle that is not a live context"
    assert b"ptrt_set_option" in P.lib.ptrt_last_error(stale)  # falls back to the thread's last message
    P.lib.ptrt_destroy(stale)  # tolerated, not freed
    assert bytes(buf.raw) == before
    # rings: same rule
    assert P.lib.ptrt_ring_map(stale, 0, ctypes.byref(ctypes.c_void_p())) == -1
    assert P.lib.ptrt_ring_unmap(stale, 0) == -1
    P.lib.ptrt_ring_destroy(stale)
    assert bytes(buf.raw) == before
    ring = ctypes.c_void_p()
    assert P.lib.ptrt_ring_create(0, 0, 2, ctypes.byref(ring)) == -1 and not ring.value


def test_reference_setters_do_not_change_what_a_frame_uses(P):
    """Scene::setSamplesPerPixel / setMaxDepth are stored and ignored by render_to_device (scene.cuh:86-87,
    1248-1255); the frame's sample count and depth are perfSettings' (scene.cuh:1044-1045)."""
    s = P.Scene(64, 64, device=P.HOST_ONLY)
    assert s.getSamplesPerPixel() == 16 and s.settings()["spp"] == 1 and s.settings()["depth"] == 4
    s.setFrameCount(5)
    s.setSamplesPerPixel(64)
    s.setMaxDepth(2)
    assert s.getSamplesPerPixel() == 64 and s.getFrameCount() == 0  # resetAccumulation
    assert s.settings()["spp"] == 1 and s.settings()["depth"] == 4
    s.setPerfSamplesPerPixel(4)
    assert s.settings()["spp"] == 4 and s.getSamplesPerPixel() == 64
    s.setPerformancePreset("ultra")  # scene.cuh:1839-1840
    assert s.settings()["spp"] == 128 and s.settings()["depth"] == 32
    s.close()


def test_farm_and_interleave_argument_checks_need_no_device(P):
    ctx, farm = ctypes.c_void_p(), ctypes.c_void_p()
    # strip `phase` of every `period`: 0 <= phase < period, and the frame must have that strip
    assert P.lib.ptrt_create_interleaved(64, 64, 3, 3, 0, ctypes.byref(ctx)) == -1 and not ctx.value
    assert P.lib.ptrt_create_interleaved(64, 20, 3, 8, 0, ctypes.byref(ctx)) == -1   # 20 rows = strips 0..2
    assert P.lib.ptrt_create_interleaved(0, 64, 0, 2, 0, ctypes.byref(ctx)) == -1
    assert P.lib.ptrt_farm_create(None, 0, ctypes.byref(farm)) == -1 and not farm.value
    buf = ctypes.create_string_buffer(256)
    dead = (ctypes.c_void_p * 1)(ctypes.cast(buf, ctypes.c_void_p))
    assert P.lib.ptrt_farm_create(dead, 1, ctypes.byref(farm)) == -1 and b"not a live context" in P.lib.ptrt_last_error(None)
    assert P.lib.ptrt_farm_gather(ctypes.cast(buf, ctypes.c_void_p), None, 0) == -1
    assert P.lib.ptrt_farm_bands(ctypes.cast(buf, ctypes.c_void_p)) == 0
    P.lib.ptrt_farm_destroy(ctypes.cast(buf, ctypes.c_void_p))
    # host-only scenes know their rows without a back end
    s = P.Scene(64, 52, device=P.HOST_ONLY, interleave=(0, 3))
    assert s.tile_rows == 8 + 8 + 4  # strips 0, 3, 6 (the last one short)
    s.close()

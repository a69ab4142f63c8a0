"""Presentation ring: the rtgl:: frame loop over pinned host memory, headless."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make(P, W=96, H=64):
    s = P.Scene(W, H)
    P.scenes.cornell(s)
    s.setPerfSamplesPerPixel(1)
    s.setMaxBounceDepth(3)
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    s.initBlueNoise()
    s.uploadToGPU()
    return s


@pytest.mark.parametrize("slots", [1, 2, 3])
def test_viewer_loop_presents_every_frame_in_order(P, tmp_path, slots):
    a = make(P)
    want = np.stack([a.render_to_host().copy() for _ in range(5)])          # the frames, one synchronous copy each
    b = make(P)
    prefix = str(tmp_path / "frame_")
    got, ms = b.view_run(5, slots=slots, dump_prefix=prefix, dump_every=2)
    assert np.array_equal(got, want) and ms > 0
    # headless draw_interop: frames 0, 2, 4 as binary PPM, top-down
    files = sorted(os.listdir(tmp_path))
    assert files == ["frame_000000.ppm", "frame_000002.ppm", "frame_000004.ppm"]
    raw = open(tmp_path / files[1], "rb").read()
    head = b"P6\n96 64\n255\n"
    assert raw.startswith(head)
    img = np.frombuffer(raw[len(head):], np.uint8).reshape(64, 96, 3)
    assert np.array_equal(img, want[2][::-1])
    a.close()
    b.close()


def test_present_ring_api_errors_and_reuse(P):
    s = make(P, 64, 64)
    dev, host = C.c_void_p(), C.c_void_p()
    assert P.lib.ptrt_present_map(s.ctx, 0, C.byref(dev)) == -1            # no ring yet
    assert P.lib.ptrt_present_create(s.ctx, 9) == -1
    assert P.lib.ptrt_present_create(s.ctx, 2) == 0
    assert P.lib.ptrt_present_map(s.ctx, 2, C.byref(dev)) == -1
    frames = []
    for f in range(4):                                                       # reuse slots while older frames are in flight
        assert P.lib.ptrt_present_map(s.ctx, f % 2, C.byref(dev)) == 0 and dev.value
        s.render_to_device(dev.value)
        assert P.lib.ptrt_present_unmap(s.ctx, f % 2) == 0
        assert P.lib.ptrt_present_acquire(s.ctx, f % 2, C.byref(host)) == 0
        frames.append(np.ctypeslib.as_array(C.cast(host, C.POINTER(C.c_uint8)), (64, 64, 3)).copy())
    t = make(P, 64, 64)
    for f in range(4):
        assert np.array_equal(frames[f], t.render_to_host())
    assert P.lib.ptrt_present_destroy(s.ctx) == 0 and P.lib.ptrt_present_destroy(s.ctx) == 0
    s.close()
    t.close()

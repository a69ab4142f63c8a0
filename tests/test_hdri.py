"""HDRI sky (Scene::loadHDRI + sampleSky's equirect lookup).
PARITY UNPINNED: the CUDA texture unit's filtering is restated from the programming guide (the
rounding of its 8-bit weights is our choice), atan2f/acosf are the deterministic versions."""
import ctypes as C
import os
import struct

import numpy as np
import pytest


def ulps(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    spacing = np.spacing(np.abs(b).astype(np.float32)).astype(np.float64)
    return np.abs(a - b) / spacing


def test_det_atan2_and_acos_accuracy(O):
    rs = np.random.RandomState(11)
    y = np.concatenate([rs.normal(0, 1, 200000), rs.normal(0, 1e-3, 20000), rs.normal(0, 1e3, 20000)]).astype(np.float32)
    x = np.concatenate([rs.normal(0, 1, 200000), rs.normal(0, 1, 20000), rs.normal(0, 1e-2, 20000)]).astype(np.float32)
    got = O.detmath(5, y, x)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert ulps(got, want).max() <= 4.0          # Cephes atanf after an fp32 quotient: 3.2 ulp worst case seen
    c = np.concatenate([rs.uniform(-1, 1, 200000), np.array([-1, 1, 0, 0.5, -0.5, 0.999999, -0.999999])]).astype(np.float32)
    got = O.detmath(6, c)
    assert ulps(got, np.arccos(c.astype(np.float64))).max() <= 3.0
    # signs, zeros, axes: what atan2f gives
    ys = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 1.0, -1.0, 0.0], np.float32)
    xs = np.array([1.0, 1.0, -1.0, -1.0, 0.0, 0.0, -0.0, -0.0, 0.0], np.float32)
    got = O.detmath(5, ys, xs)
    want = np.arctan2(ys, xs).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def rgbe_encode(rgb):
    """(h, w, 3) float -> (h, w, 4) uint8 Radiance RGBE, and the floats a reader must return."""
    m = rgb.max(axis=2)
    e = np.where(m > 1e-32, np.floor(np.log2(np.maximum(m, 1e-38))).astype(np.int32) + 1, -200)
    scale = np.where(m > 1e-32, np.ldexp(256.0, -e), 0.0)
    q = np.clip(np.floor(rgb * scale[..., None]), 0, 255).astype(np.uint8)
    ebyte = np.where(m > 1e-32, e + 128, 0).astype(np.uint8)
    out = np.concatenate([q, ebyte[..., None]], axis=2)
    dec = np.where(ebyte[..., None] != 0, q.astype(np.float32) * np.ldexp(np.float32(1.0), ebyte.astype(np.int32) - 136)[..., None],
                   np.float32(0.0)).astype(np.float32)
    return out, dec


def write_hdr(path, rgbe, rle):
    h, w, _ = rgbe.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\n# written by tests/test_hdri.py\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n")
        f.write(f"-Y {h} +X {w}\n".encode())
        for y in range(h):
            if not rle:
                f.write(rgbe[y].tobytes())
                continue
            f.write(struct.pack("BBBB", 2, 2, w >> 8, w & 255))
            for ch in range(4):
                row = rgbe[y, :, ch]
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, row[x]]))
                        x += run
                    else:
                        lit = min(w - x, 100)
                        f.write(bytes([lit]) + row[x:x + lit].tobytes())
                        x += lit


@pytest.mark.parametrize("rle", [False, True])
def test_radiance_reader_matches_the_format(P, tmp_path, rle):
    rs = np.random.RandomState(4)
    rgb = (rs.uniform(0, 1, (12, 40, 3)) ** 4 * 50).astype(np.float32)
    rgb[3, 5:30] = 2.0          # a long run for the RLE path
    rgb[7, :4] = 0.0            # exponent byte 0 -> exact zero
    rgbe, dec = rgbe_encode(rgb)
    path = tmp_path / "sky.hdr"
    write_hdr(path, rgbe, rle)
    s = P.Scene(32, 32, device=P.HOST_ONLY)
    P.scenes.cornell(s)
    s.loadHDRI(path)
    d = s.flatten().contents
    assert (d.env_width, d.env_height) == (40, 12) and d.use_sky == 1
    got = np.ctypeslib.as_array(d.env_rgba, (12, 40, 4))
    assert np.array_equal(got[..., :3], dec[::-1]) and (got[..., 3] == 1.0).all()      # bottom row first (vertical flip)
    assert np.abs(got[::-1, :, :3] - rgb).max() <= np.abs(rgb).max() / 128 + 1e-6      # and it is the picture, to RGBE precision
    s.freeHDRI()
    assert not s.flatten().contents.env_rgba
    with pytest.raises(P.PtrtError, match="Failed to load HDRI"):
        s.loadHDRI(tmp_path / "missing.hdr")
    bad = tmp_path / "bad.hdr"
    bad.write_bytes(b"P6\n1 1\n255\n\0\0\0")
    with pytest.raises(P.PtrtError, match="not a Radiance picture"):
        s.loadHDRI(bad)


def far_cube_scene(P, W, H, **kw):
    s = P.Scene(W, H, **kw)
    m = s.addCube(P.Material((0.7, 0.7, 0.7), 0.5))
    s.scale(m, (0.2, 0.2, 0.2))
    s.moveTo(m, (0.0, -3.0, -30.0))
    s.setCamera((0, 0, 0), (0, 0, -1), (0, 1, 0), 70.0)
    return s


def test_oracle_env_sky_orientation_and_constant_map(P, O, blue_noise):
    W, H = 48, 32
    s = far_cube_scene(P, W, H, device=P.HOST_ONLY)
    env = np.zeros((16, 32, 4), np.float32)
    env[..., :3] = (0.25, 0.5, 0.75)
    env[..., 3] = 1.0
    s.setEnvironmentMap(env)
    r = O.render(s.flatten(), W, H, 1, 2, 0, blue_noise, O.xorwow_init(1, 0, W * H))
    sky = r["object_id"] < 0
    assert sky.mean() > 0.95
    assert np.allclose(r["accum"][sky], (0.25, 0.5, 0.75), rtol=3e-7)      # a constant map is a constant sky
    # v = acos(dir.y)/pi: looking up reads the first rows, looking down the last ones; u wraps around in x
    env[:8, :, :3] = (1.0, 0.0, 0.0)
    env[8:, :, :3] = (0.0, 0.0, 1.0)
    s.setEnvironmentMap(env)
    img = O.render(s.flatten(), W, H, 1, 2, 0, blue_noise, O.xorwow_init(1, 0, W * H))["accum"].reshape(H, W, 3)
    assert img[2, :, 0].min() > 0.9 and img[2, :, 2].max() < 0.1           # top of the view: red half
    assert img[-3, :4, 2].min() > 0.9 and img[-3, :4, 0].max() < 0.1       # bottom: blue half
    # -z is the view direction: phi = atan2(-1, 0) = -pi/2 -> u = 0.25: a stripe there shows up in the view centre
    env[..., :3] = 0.0
    env[:, 7:9, :3] = 5.0
    s.setEnvironmentMap(env)
    img = O.render(s.flatten(), W, H, 1, 2, 0, blue_noise, O.xorwow_init(1, 0, W * H))["accum"].reshape(H, W, 3)
    cols = img[H // 2 - 6, :, 0]
    assert cols[W // 2 - 2: W // 2 + 2].min() > 2.0 and cols[:6].max() == 0.0 and cols[-6:].max() == 0.0


@pytest.mark.gpu
def test_gpu_det_atan2_acos_bits(P, O):
    s = P.Scene(16, 16)
    rs = np.random.RandomState(2)
    y = np.concatenate([rs.normal(0, 1, 1 << 16), [0.0, -0.0, 0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan]]).astype(np.float32)
    x = np.concatenate([rs.normal(0, 1, 1 << 16), [1.0, -1.0, -0.0, 0.0, 0.0, -0.0, np.inf, 3.0, 1.0]]).astype(np.float32)
    out = np.zeros_like(y)
    fp = C.POINTER(C.c_float)
    assert P.lib.ptrt_debug_detmath(s.ctx, 5, y.ctypes.data_as(fp), x.ctypes.data_as(fp), y.size, out.ctypes.data_as(fp)) == 0
    assert np.array_equal(out.view(np.uint32), O.detmath(5, y, x).view(np.uint32))
    c = np.concatenate([rs.uniform(-1, 1, 1 << 16), [-1.0, 1.0, 0.5, -0.5, 0.0, 1.5, np.nan]]).astype(np.float32)
    out = np.zeros_like(c)
    assert P.lib.ptrt_debug_detmath(s.ctx, 6, c.ctypes.data_as(fp), None, c.size, out.ctypes.data_as(fp)) == 0
    assert np.array_equal(out.view(np.uint32), O.detmath(6, c).view(np.uint32))
    s.close()


@pytest.mark.gpu
def test_gpu_env_sky_bit_exact(P, O, blue_noise):
    from common import assert_frames_equal, render_both
    rs = np.random.RandomState(9)
    env = np.ones((37, 64, 4), np.float32)
    env[..., :3] = rs.uniform(0, 1, (37, 64, 3)) ** 6 * 30          # a few hot texels, odd height
    s = P.Scene(96, 64)
    P.scenes.showcase(s, segments=10)                               # glass / metal spheres under an open sky
    s.setEnvironmentMap(env)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 5, 2)
    assert_frames_equal(gpu, cpu)
    assert (gpu[0]["object_id"] < 0).mean() > 0.05                  # part of the sky is seen directly
    base = gpu[1]["accum"].copy()
    s.freeHDRI()                                                    # back to the gradient
    rgb = s.render_to_host()
    assert not np.array_equal(s.read(P.BUF_ACCUM), base)
    t = far_cube_scene(P, 64, 64)
    t.setEnvironmentMap(env)
    gpu, cpu = render_both(P, O, t, blue_noise, 1, 3, 1)
    assert_frames_equal(gpu, cpu)
    s.close()
    t.close()

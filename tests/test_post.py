"""Bloom chain + resolution scale: oracle properties on CPU, GPU parity (bit-exact)
on the box.  PARITY UNPINNED w.r.t. the CUDA reference (no vectors exist for this stage)."""
import numpy as np
import pytest


def cornell(P, w, h, **kw):
    s = P.Scene(w, h, **kw)
    P.scenes.cornell(s)
    return s


# ------------------------------------------------------------------------------------------ CPU
def test_oracle_bloom_properties(O):
    W, H = 128, 128
    n = W * H
    dim = np.full((n, 3), 0.45, np.float32)                   # weight clamp((b - 1.5 + 0.5)/1 + 0.5, 0, 1) = 0 up to b = 0.5
    assert np.array_equal(O.bloom(dim, W, H), dim)
    # a constant image above the threshold (weight 1) passes the bright pass unchanged; every mip level then
    # holds ~c (the 5 taps sum to 1), the up chain accumulates 6 of them and the frame gets c + 6c
    c = np.full((n, 3), 2.5, np.float32)
    out = O.bloom(c, W, H).reshape(H, W, 3)
    assert np.allclose(out, 7 * 2.5, rtol=1e-4)
    # bloom only ever adds light, and a bright spot spreads symmetrically around its pixel block
    rs = np.random.RandomState(3)
    img = rs.uniform(0, 0.45, (H, W, 3)).astype(np.float32)
    img[62:66, 62:66] = 40.0
    out = O.bloom(img.reshape(-1, 3), W, H).reshape(H, W, 3)
    assert (out >= img).all() and out[40, 64, 0] > img[40, 64, 0] + 1e-3 and out[64, 20, 0] > img[64, 20, 0]
    glow = (out - img)[..., 0]
    assert glow[64, 64] > glow[64, 90] > glow[64, 120] >= 0
    # odd sizes follow the reference's doubled-size bookkeeping: rows/columns past 2*floor(n/2) get no bloom
    W2, H2 = 131, 77
    img2 = rs.uniform(0, 3.0, (H2 * W2, 3)).astype(np.float32)
    out2 = O.bloom(img2, W2, H2)
    assert np.isfinite(out2).all() and out2.shape == img2.shape
    with pytest.raises(RuntimeError):
        O.bloom(np.zeros((32 * 32, 3), np.float32), 32, 32)      # level 5 would be 0x0: the reference reads NULL


def test_oracle_upscale_properties(O):
    rs = np.random.RandomState(5)
    img = rs.uniform(0, 4, (48 * 64, 3)).astype(np.float32)
    assert np.array_equal(O.upscale(img, 64, 48, 64, 48), img)                  # same size: exact copy
    up = O.upscale(img, 128, 96, 64, 48).reshape(96, 128, 3)
    lo = img.reshape(48, 64, 3)
    assert up.min() >= lo.min() - 1e-6 and up.max() <= lo.max() + 1e-6         # convex combinations
    ramp = np.repeat(np.arange(64, dtype=np.float32)[None, :, None], 48, axis=0).repeat(3, axis=2)
    upr = O.upscale(ramp.reshape(-1, 3), 128, 96, 64, 48).reshape(96, 128, 3)
    assert np.allclose(np.diff(upr[10, 2:-2, 0]), 0.5, atol=1e-5)              # a ramp stays a ramp
    assert np.allclose(upr[:, 0, 0], 0.0) and np.allclose(upr[:, -1, 0], 63.0)  # clamped at the borders


def test_render_size_follows_resolution_scale(P):
    s = cornell(P, 1920, 1080, device=P.HOST_ONLY)
    assert s.renderSize() == (1920, 1080)
    f32 = np.float32
    for preset, scale in (("performance", 0.75), ("fast", 0.35), ("balanced", 1.0)):
        s.setPerformancePreset(preset)
        assert s.renderSize() == (int(f32(1920) * f32(scale)), int(f32(1080) * f32(scale))), preset
    s.setResolutionScale(0.1)                                   # clamped to 0.25
    assert s.renderSize() == (480, 270)
    small = cornell(P, 96, 80, device=P.HOST_ONLY)
    small.setResolutionScale(0.25)                              # never below 64 pixels
    assert small.renderSize() == (64, 64)
    tiny = cornell(P, 48, 32, device=P.HOST_ONLY)
    tiny.setResolutionScale(0.5)                                # ... nor above the frame
    assert tiny.renderSize() == (48, 32)


# ------------------------------------------------------------------------------------------ GPU
def setup(P, W, H, spp=2, depth=4, denoise=False, bloom=False, scale=1.0):
    s = cornell(P, W, H)
    s.setPerfSamplesPerPixel(spp)
    s.setMaxBounceDepth(depth)
    s.setDenoiserEnabled(denoise)
    s.setBloomEnabled(bloom)
    s.setResolutionScale(scale)
    s.initBlueNoise()
    s.uploadToGPU()
    return s


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(128, 96), (131, 77)])
def test_gpu_bloom_bit_exact(P, O, blue_noise, size):
    W, H = size
    s = setup(P, W, H, bloom=True)
    rng = O.xorwow_init(P.DEFAULT_SEED, 0, W * H)
    d = s.flatten()
    for f in range(2):
        rgb = s.render_to_host()
        r = O.render(d, W, H, 2, 4, f, blue_noise, rng, threads=8)
        want = O.bloom(r["accum"], W, H)
        assert (want > r["accum"] + 1e-4).any()                       # the ceiling light does bloom
        got = s.read(P.BUF_ACCUM)                                      # bloom is added into the colour buffer in place
        bad = np.flatnonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))
        assert bad.size == 0, f"frame {f}: {bad.size} px differ, first {bad[:5]}: {got[bad[0]]} vs {want[bad[0]]}"
        assert np.array_equal(rgb, O.tonemap(want, W, H))
        assert np.array_equal(s.read(P.BUF_RNG), rng)
    s.setBloomEnabled(False)                                           # and off again: the plain frame
    rgb = s.render_to_host()
    r = O.render(d, W, H, 2, 4, 2, blue_noise, rng, threads=8)
    assert np.array_equal(s.read(P.BUF_ACCUM).view(np.uint32), r["accum"].view(np.uint32))
    assert np.array_equal(rgb, O.tonemap(r["accum"], W, H))
    s.close()


@pytest.mark.gpu
def test_gpu_balanced_preset_denoise_then_bloom(P, O, blue_noise):
    W, H = 128, 96
    s = setup(P, W, H, spp=1, denoise=True, bloom=True)
    rng = O.xorwow_init(P.DEFAULT_SEED, 0, W * H)
    dn = O.Denoiser(W, H)
    d = s.flatten()
    for f in range(3):
        pvp = s.view_proj()
        rgb = s.render_to_host()
        r = O.render(d, W, H, 1, 4, f, blue_noise, rng, threads=8)
        mv = O.motion_vectors(r["depth"], W, H, d.contents.camera, pvp)
        den = dn.denoise(r["accum"], r["normal"], r["depth"], mv, r["object_id"])
        want = O.bloom(den, W, H)                                      # bloom goes INTO the denoised image
        assert np.array_equal(s.read(P.BUF_ACCUM).view(np.uint32), r["accum"].view(np.uint32)), f
        assert np.array_equal(s.read(P.BUF_DENOISED).view(np.uint32), want.view(np.uint32)), f
        assert np.array_equal(rgb, O.tonemap(want, W, H)), f
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("denoise,bloom", [(False, False), (True, False), (True, True)])
def test_gpu_resolution_scale_bit_exact(P, O, blue_noise, denoise, bloom):
    W, H = 192, 128
    s = setup(P, W, H, spp=1, denoise=denoise, bloom=bloom, scale=0.5)
    rw, rh = s.renderSize()
    assert (rw, rh) == (96, 64)
    rng_full = O.xorwow_init(P.DEFAULT_SEED, 0, W * H)
    rng = rng_full[: rw * rh].copy()        # pixel p of the small frame continues state p (scene.cuh:1091-1098)
    dn = O.Denoiser(rw, rh)
    d = s.flatten()
    for f in range(3):
        pvp = s.view_proj()
        rgb = s.render_to_host()
        r = O.render(d, rw, rh, 1, 4, f, blue_noise, rng, threads=8)
        assert np.array_equal(s.read(P.BUF_RENDER_ACCUM).view(np.uint32), r["accum"].view(np.uint32)), f
        assert np.array_equal(s.read(P.BUF_OBJECT_ID), r["object_id"]), f
        assert np.array_equal(s.read(P.BUF_DEPTH).view(np.uint32), r["depth"].view(np.uint32)), f
        cur = r["accum"]
        if denoise:
            mv = O.motion_vectors(r["depth"], rw, rh, d.contents.camera, pvp)
            assert np.array_equal(s.read(P.BUF_MOTION).view(np.uint32), mv.view(np.uint32)), f
            cur = dn.denoise(r["accum"], r["normal"], r["depth"], mv, r["object_id"])
        if bloom:
            cur = O.bloom(cur, rw, rh, W, H)
        if denoise:
            assert np.array_equal(s.read(P.BUF_DENOISED).view(np.uint32), cur.view(np.uint32)), f
        final = O.upscale(cur, W, H, rw, rh)
        got = s.read(P.BUF_ACCUM)
        bad = np.flatnonzero((got.view(np.uint32) != final.view(np.uint32)).any(axis=1))
        assert bad.size == 0, f"frame {f}: {bad.size} px differ, first {bad[:5]}: {got[bad[0]]} vs {final[bad[0]]}"
        assert np.array_equal(rgb, O.tonemap(final, W, H)), f
        state = s.read(P.BUF_RNG)
        assert np.array_equal(state[: rw * rh], rng) and np.array_equal(state[rw * rh:], rng_full[rw * rh:])
    # back to full size: buffers are re-created, accumulation restarts at frame 0
    s.setResolutionScale(1.0)
    assert s.renderSize() == (W, H) and s.getFrameCount() == 0
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    rng_full[: rw * rh] = rng
    rgb = s.render_to_host()
    r = O.render(d, W, H, 1, 4, 0, blue_noise, rng_full, threads=8)
    assert np.array_equal(s.read(P.BUF_ACCUM).view(np.uint32), r["accum"].view(np.uint32))
    assert np.array_equal(rgb, O.tonemap(r["accum"], W, H))
    s.close()


@pytest.mark.gpu
def test_post_chain_needs_a_full_frame_of_64_pixels(P):
    band = P.Scene(128, 128, tile_y0=32, tile_rows=32)
    assert P.lib.ptrt_set_bloom(band.ctx, 1) == -1 and b"full-frame" in P.lib.ptrt_last_error(band.ctx)
    assert P.lib.ptrt_set_render_size(band.ctx, 64, 64) == -1
    band.close()
    small = P.Scene(48, 48)
    assert P.lib.ptrt_set_bloom(small.ctx, 1) == -1 and b"64x64" in P.lib.ptrt_last_error(small.ctx)
    assert P.lib.ptrt_set_render_size(small.ctx, 49, 48) == -1
    small.close()

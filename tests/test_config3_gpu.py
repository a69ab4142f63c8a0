"""BASELINE.json configs[3]: the showcase scene at 3840x2160, 8 spp, split into 8 horizontal bands (one per GPU of a
node, gathered over RCCL).  The 8-GPU run belongs to the driver; what one GPU can pin is everything but the
transport: at the full size the default traversal equals the plain one buffer for buffer, eight band contexts (270
rows each, what the eight ranks render) concatenate to the bytes of the full-frame context, and at a small size the
eight bands equal the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W4K, H4K, SPP, DEPTH, BANDS = 3840, 2160, 8, 4, 8
KINDS = ("accum", "normal", "depth", "object_id", "rng")


def _setup(P, s, spp=SPP, depth=DEPTH, opts=None):
    P.scenes.showcase(s)
    s.setPerfSamplesPerPixel(spp)
    s.setMaxBounceDepth(depth)
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    s.initBlueNoise()
    s.uploadToGPU()
    s.set_option("count_rays", 1)
    for k, v in (opts or {}).items():
        s.set_option(k, v)


def _frame(P, s):
    rgb = s.render_to_host()
    out = dict(rgb8=rgb, stats=s.stats())
    for k, b in zip(KINDS, (P.BUF_ACCUM, P.BUF_NORMAL, P.BUF_DEPTH, P.BUF_OBJECT_ID, P.BUF_RNG)):
        a = s.read(b)
        out[k] = a.view(np.uint32) if a.dtype == np.float32 else a
    return out


def test_showcase_4k_8spp_variants_agree_and_eight_bands_equal_the_frame(P):
    from ptrt_amd import tilefarm
    full = P.Scene(W4K, H4K)
    _setup(P, full)
    ref = _frame(P, full)
    full.close()
    n = W4K * H4K
    assert ref["stats"]["paths"] == SPP * n and ref["accum"].any() and (ref["object_id"] >= -1).all()
    # the plain traversal (batches of 64 pairs, every lane its own leaf, no stealing) and the merged one
    for opts in (dict(fetch_min=0, leaf_pairs=0, steal=0, leaf_min=64), dict(merged=1)):
        s = P.Scene(W4K, H4K)
        _setup(P, s, opts=opts)
        got = _frame(P, s)
        s.close()
        for k in KINDS + ("rgb8",):
            assert np.array_equal(ref[k], got[k]), f"{opts}: {k} differs in {(ref[k] != got[k]).sum()} words"
        assert ref["stats"] == got["stats"]
    # eight band contexts on this GPU = the eight ranks of configs[3]
    parts, rays = [], 0
    for y0, rows in tilefarm.bands(H4K, BANDS):
        assert rows == 270
        b = P.Scene(W4K, H4K, tile_y0=y0, tile_rows=rows)
        _setup(P, b)
        parts.append(_frame(P, b))
        rays += parts[-1]["stats"]["extension_rays"] + parts[-1]["stats"]["shadow_rays"]
        b.close()
    for k in KINDS:  # HDR image, G-buffers and generator states are top-down: plain concatenation
        assert np.array_equal(np.concatenate([p[k] for p in parts], axis=0), ref[k]), k
    # the RGB8 image is bottom-up (scene.cuh:2013-2015): the bands stack in reverse rank order, as tilefarm.frame_views lays them out
    assert np.array_equal(np.concatenate([p["rgb8"] for p in reversed(parts)], axis=0), ref["rgb8"])
    assert rays == ref["stats"]["extension_rays"] + ref["stats"]["shadow_rays"]
    # band load balance of this frame (what the 8-GPU strong scaling is limited by)
    per = [p["stats"]["extension_rays"] + p["stats"]["shadow_rays"] for p in parts]
    print("rays per band:", per, "max/mean", max(per) / (sum(per) / len(per)))


def test_eight_bands_of_a_small_showcase_frame_equal_the_oracle(P, O, blue_noise):
    from common import assert_frames_equal, render_both
    from ptrt_amd import tilefarm
    W, H = 96, 64
    for y0, rows in tilefarm.bands(H, BANDS):
        b = P.Scene(W, H, tile_y0=y0, tile_rows=rows)
        P.scenes.showcase(b, segments=12)
        gpu, cpu = render_both(P, O, b, blue_noise, SPP, DEPTH, frames=2)
        assert_frames_equal(gpu, cpu)
        b.close()


def test_showcase_4k_8spp_frame_equals_the_oracle(P, O, blue_noise):
    """configs[3]'s frame at its full size against the oracle (16 host threads, ~10 s): every buffer, generator state, ray count."""
    from common import assert_frames_equal, render_both
    s = P.Scene(W4K, H4K)
    P.scenes.showcase(s)
    gpu, cpu = render_both(P, O, s, blue_noise, SPP, DEPTH, 1, threads=16)
    assert_frames_equal(gpu, cpu)
    s.close()

"""Scene::setPerformancePreset (scene.cuh:1833-1879) end to end on the GPU, against the oracle.  `ultra` -- 128 spp,
32 bounces (written past setMaxBounceDepth's clamp of 16), bloom on, denoiser off -- is the heaviest setting a caller
of the reference can reach; the other presets differ in bounce depth, render size and post chain (tests/test_post.py,
tests/test_denoiser.py cover those stages)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ultra_preset_128spp_32_bounces_with_bloom(P, O, blue_noise):
    W, H = 64, 64  # the bloom chain needs six non-empty mip levels
    s = P.Scene(W, H)
    P.scenes.showcase(s, segments=10)
    s.setPerformancePreset("ultra")
    st = s.settings()
    assert (st["spp"], st["depth"], st["denoiser"], st["bloom"], st["scale"]) == (128, 32, False, True, 1.0)
    s.initBlueNoise()
    s.uploadToGPU()
    s.set_option("count_rays", 1)
    rng = O.xorwow_init(P.DEFAULT_SEED, 0, W * H)
    d = s.flatten()
    for f in range(2):
        rgb = s.render_to_host()
        r = O.render(d, W, H, 128, 32, f, blue_noise, rng, threads=16)
        want = O.bloom(r["accum"], W, H)
        got = s.read(P.BUF_ACCUM)  # bloom is added into the colour buffer in place
        bad = np.flatnonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))
        assert bad.size == 0, f"frame {f}: {bad.size} px differ, first {bad[:5]}: {got[bad[0]]} vs {want[bad[0]]}"
        assert np.array_equal(rgb, O.tonemap(want, W, H))
        assert np.array_equal(s.read(P.BUF_RNG), rng)
        assert np.array_equal(s.read(P.BUF_OBJECT_ID), r["object_id"])
        assert s.stats() == r["stats"]
        # deep paths do occur: more than 4 rays per path somewhere means bounces past the default depth were taken
        assert r["stats"]["extension_rays"] > 1.5 * r["stats"]["paths"]
    s.close()


@pytest.mark.parametrize("preset", ["fast", "performance", "balanced", "quality"])
def test_presets_render_and_keep_their_settings(P, preset):
    """The published-workload scene class (large spheres, single-leaf TLAS) under every real-time preset: the frame
    renders, the settings are the reference's, and two identical scenes give identical bytes."""
    want = {"fast": (2, False, False, 0.35), "performance": (3, True, False, 0.75), "balanced": (4, True, True, 1.0),
            "quality": (6, True, True, 1.0)}[preset]
    frames = []
    for _ in range(2):
        s = P.Scene(320, 192)
        P.scenes.million(s, segments=24)
        s.setPerformancePreset(preset)
        st = s.settings()
        assert (st["depth"], st["denoiser"], st["bloom"]) == want[:3] and abs(st["scale"] - want[3]) < 1e-6 and st["spp"] == 1
        s.initBlueNoise()
        s.uploadToGPU()
        frames.append([s.render_to_host() for _ in range(3)])
        s.close()
    for a, b in zip(*frames):
        assert a.any() and np.array_equal(a, b)

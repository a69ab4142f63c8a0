"""The CPU oracle on its own: behavioural facts recorded for the reference,
size-independent properties, and regression goldens of the oracle's own output
(tests/golden/make_oracle_golden.py).  PARITY UNPINNED w.r.t. the CUDA reference."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def cornell(P, w, h, **kw):
    s = P.Scene(w, h, device=P.HOST_ONLY, **kw)
    P.scenes.cornell(s)
    return s


def test_depth1_is_black_one_ray_per_pixel(P, O, blue_noise):
    """Primary rays are flagged specular -> no NEE at the first hit; the emitter is out of view
    (camera.cuh:204, path_logic.cuh:840): max_depth 1 renders black with exactly 1 ray per pixel."""
    s = cornell(P, 64, 64)
    rng = O.xorwow_init(12345, 0, 64 * 64)
    r = O.render(s.flatten(), 64, 64, 1, 1, 0, blue_noise, rng, threads=4)
    assert not r["accum"].any()
    assert r["stats"] == dict(extension_rays=4096, shadow_rays=0, paths=4096, shadow_rays_walked=0)
    assert (r["object_id"] >= 0).all() and set(np.unique(r["object_id"])) <= set(range(8))
    assert np.all(r["depth"] > 4.9) and np.all(r["depth"] < 17)
    assert np.allclose(np.linalg.norm(r["normal"], axis=1), 1.0, atol=1e-6)
    assert not O.tonemap(r["accum"], 64, 64).any()


def test_depth4_ray_budget_and_draw_count(P, O, blue_noise):
    s = cornell(P, 96, 96)
    n = 96 * 96
    rng0 = O.xorwow_init(12345, 0, n)
    rng = rng0.copy()
    r = O.render(s.flatten(), 96, 96, 1, 4, 0, blue_noise, rng, threads=8)
    rays = (r["stats"]["extension_rays"] + r["stats"]["shadow_rays"]) / n
    assert 4.5 < rays < 5.3                       # SURVEY probe: 4.91 rays per pixel-sample
    assert r["accum"].max() <= 100.0 * 3 and r["accum"].min() >= 0.0 and np.isfinite(r["accum"]).all()
    draws = ((rng[:, 0].astype(np.int64) - rng0[:, 0]) % (1 << 32)) // 362437   # d is a Weyl counter
    assert draws.min() >= 3 and draws.max() <= 17  # Appendix C: at most 17 uniforms per sample


def test_thread_count_and_tiling_do_not_change_bits(P, O, blue_noise):
    s = cornell(P, 80, 48)
    d = s.flatten()
    a = O.render(d, 80, 48, 2, 4, 3, blue_noise, O.xorwow_init(12345, 0, 80 * 48), threads=1)
    b = O.render(d, 80, 48, 2, 4, 3, blue_noise, O.xorwow_init(12345, 0, 80 * 48), threads=7)
    assert np.array_equal(a["accum"].view(np.uint32), b["accum"].view(np.uint32))
    t = O.render(d, 80, 48, 2, 4, 3, blue_noise, O.xorwow_init(12345, 16 * 80, 8 * 80), tile_y0=16, tile_rows=8)
    assert np.array_equal(t["accum"].view(np.uint32), a["accum"][16 * 80:24 * 80].view(np.uint32))
    assert np.array_equal(t["object_id"], a["object_id"][16 * 80:24 * 80])


def test_frames_continue_the_random_streams(P, O, blue_noise):
    """Per-pixel states persist across frames (scene_kernels.cuh:188); frame index only moves the jitter."""
    s = cornell(P, 48, 48)
    d = s.flatten()
    rng = O.xorwow_init(12345, 0, 48 * 48)
    f0 = O.render(d, 48, 48, 1, 4, 0, blue_noise, rng)
    f1 = O.render(d, 48, 48, 1, 4, 1, blue_noise, rng)
    assert not np.array_equal(f0["accum"], f1["accum"])
    # two 1-spp frames from one stream == what a 2-spp frame consumes, sample by sample
    rng2 = O.xorwow_init(12345, 0, 48 * 48)
    both = O.render(d, 48, 48, 2, 4, 0, blue_noise, rng2)
    assert np.array_equal(rng, rng2)
    assert np.allclose(both["accum"], (f0["accum"] + f1["accum"]) / 2, rtol=1e-6, atol=1e-7)
    assert np.array_equal(both["object_id"], f0["object_id"])


def test_bvh_topology_does_not_change_first_hits(P, O):
    """Closest hits are topology independent except for exact ties."""
    rs = np.random.RandomState(3)
    n = 20000
    o = np.array([0, 0, -5], np.float32) + rs.uniform(-4, 4, (n, 3)).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hits = []
    for leaf in ((12, 5), (2, 0), (1, 0)):
        s = P.Scene(32, 32, device=P.HOST_ONLY)
        P.scenes.showcase(s, segments=8)
        s.setBVHLeafTarget(*leaf)
        hits.append(O.trace_rays(s.flatten(), o, d))
    for h in hits[1:]:
        same = (h["mesh_index"] == hits[0]["mesh_index"]) & (h["t"] == hits[0]["t"])
        assert same.mean() > 0.9995
    assert hits[0]["hit"].mean() > 0.5
    # shadow query agrees with the closest hit wherever no transmissive mesh is involved
    s = P.Scene(32, 32, device=P.HOST_ONLY)
    P.scenes.cornell(s)
    h = O.trace_rays(s.flatten(), o, d)
    occ = O.any_hit(s.flatten(), o, d, np.full(n, 1e30, np.float32))
    assert np.array_equal(occ.astype(bool), h["hit"].astype(bool))
    occ_short = O.any_hit(s.flatten(), o, d, (h["t"] * 0.5).astype(np.float32))
    assert not occ_short[h["hit"] == 1].any()


def test_tonemap_known_points(O):
    acc = np.array([[0, 0, 0], [1e9, 1e9, 1e9], [0.18, 0.18, 0.18], [1.0, 0.0, 0.0]], np.float32)
    rgb = O.tonemap(acc, 4, 1)
    assert rgb[0, 0].tolist() == [0, 0, 0] and rgb[0, 1].tolist() == [255, 255, 255]
    assert 80 <= rgb[0, 2, 0] <= 140 and rgb[0, 2, 0] == rgb[0, 2, 1] == rgb[0, 2, 2]  # ACES fit of mid grey
    assert rgb[0, 3, 0] > 200 and rgb[0, 3, 1] < rgb[0, 3, 0]
    two = O.tonemap(np.array([[1, 1, 1], [0, 0, 0]], np.float32).reshape(2, 1, 3).reshape(-1, 3), 1, 2)
    assert two[0, 0, 0] == 0 and two[1, 0, 0] > 0                                       # rows flipped


@pytest.mark.parametrize("name", ["cornell_64x64_1spp_d4_f0", "cornell_64x64_4spp_d2_f3", "showcase12_64x48_2spp_d5_f0"])
def test_oracle_regression_goldens(P, O, blue_noise, name):
    g = np.load(os.path.join(GOLD, f"oracle_{name}.npz"))
    import make_oracle_golden as M
    r = M.render_case(P, O, blue_noise, name)
    assert np.array_equal(r["object_id"], g["object_id"])
    assert np.array_equal(r["accum"].view(np.uint32), g["accum"].view(np.uint32))
    assert np.array_equal(r["depth"].view(np.uint32), g["depth"].view(np.uint32))
    assert np.array_equal(r["rgb8"], g["rgb8"])

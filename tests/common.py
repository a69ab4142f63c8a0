"""Helpers shared by the parity tests: build a scene twice (GPU-backed and host-only is not
needed: one Scene serves both, the oracle reads its flattened arrays), render with the HIP
back end and with the CPU oracle, compare every buffer bit for bit."""
import numpy as np


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def render_both(P, O, scene, blue_noise, spp, depth, frames=1, threads=8, count=True):
    """Renders `frames` consecutive frames on the GPU and in the oracle from the same initial
    generator states.  Returns (gpu, cpu): lists of dicts of buffers per frame."""
    W, rows, y0 = scene.width, scene.tile_rows, scene.tile_y0
    scene.setPerfSamplesPerPixel(spp)
    scene.setMaxBounceDepth(depth)
    scene.setDenoiserEnabled(False)
    scene.setBloomEnabled(False)
    scene.initBlueNoise()
    scene.uploadToGPU()
    scene.reset_rng(P.DEFAULT_SEED)
    if count:
        scene.set_option("count_rays", 1)
    desc = scene.flatten()
    rng = O.xorwow_init(P.DEFAULT_SEED, y0 * W, rows * W)
    gpu_rng0 = scene.read(P.BUF_RNG)
    assert np.array_equal(gpu_rng0, rng), "XORWOW initial states differ"
    gpu, cpu = [], []
    for f in range(frames):
        assert scene.getFrameCount() == f
        rgb = scene.render_to_host()
        g = dict(accum=scene.read(P.BUF_ACCUM), normal=scene.read(P.BUF_NORMAL), depth=scene.read(P.BUF_DEPTH),
                 object_id=scene.read(P.BUF_OBJECT_ID), rgb8=rgb, rng=scene.read(P.BUF_RNG))
        if count:
            g["stats"] = scene.stats()
        gpu.append(g)
        c = O.render(desc, W, scene.height, spp, depth, f, blue_noise, rng, tile_y0=y0, tile_rows=rows,
                     threads=threads)
        c["rgb8"] = O.tonemap(c["accum"], W, rows, threads=threads)
        c["rng"] = rng.copy()
        cpu.append(c)
    return gpu, cpu


def assert_frames_equal(gpu, cpu, check_stats=True, walks_all=False):
    """`walks_all`: the kernel shape under test walks every shadow ray (the asynchronous-lane kernel), so its
    shadow_rays_walked equals shadow_rays instead of the oracle's count without the zero-valued light samples."""
    for f, (g, c) in enumerate(zip(gpu, cpu)):
        assert np.array_equal(g["object_id"], c["object_id"]), f"frame {f}: objectId differs at " \
            f"{np.flatnonzero(g['object_id'] != c['object_id'])[:8]}"
        for k in ("depth", "normal", "accum"):
            gb, cb = bits(g[k]), bits(c[k])
            bad = np.flatnonzero((gb != cb).reshape(gb.shape[0], -1).any(axis=1))
            assert bad.size == 0, f"frame {f}: {k} differs in {bad.size} pixels, first {bad[:8]}; " \
                f"gpu={g[k][bad[0]]} cpu={c[k][bad[0]]}"
        assert np.array_equal(g["rng"], c["rng"]), f"frame {f}: generator states differ"
        assert np.array_equal(g["rgb8"], c["rgb8"]), f"frame {f}: RGB8 differs"
        if check_stats and "stats" in g:
            want = dict(c["stats"], shadow_rays_walked=c["stats"]["shadow_rays"]) if walks_all else c["stats"]
            assert g["stats"] == want, f"frame {f}: ray counts {g['stats']} vs {want}"

"""Wavefront stages (csrc/pt_wavefront.hip.h: shade / persistent trace over the whole frame's rays)
against the CPU oracle, bit for bit -- accum, normal, depth, objectId, RGB8, generator states and
ray counts -- on every kind of scene the stages accept, and against the megakernel at a size where
the oracle would be slow."""
import numpy as np
import pytest

from common import assert_frames_equal, bits, render_both

pytestmark = pytest.mark.gpu


def mode(P, s):
    import ctypes
    P.lib.ptrt_debug_last_render_mode.argtypes = [ctypes.c_void_p]
    return P.lib.ptrt_debug_last_render_mode(s.ctx)


@pytest.mark.parametrize("size,spp,depth,frames", [((64, 64), 1, 1, 1), ((96, 72), 4, 4, 3), ((61, 45), 2, 3, 2)])
def test_cornell_single_leaf_blases(P, O, blue_noise, size, spp, depth, frames):
    s = P.Scene(size[0], size[1])
    P.scenes.cornell(s)
    s.set_option("wavefront", 1)
    s.set_option("wf_sort", 1 if spp == 2 else 0)  # (a ragged frame through the sorting shade stage as well)
    gpu, cpu = render_both(P, O, s, blue_noise, spp, depth, frames)
    assert mode(P, s) == 1
    assert_frames_equal(gpu, cpu)
    s.close()


@pytest.mark.parametrize("fetch_min,wf_sort", [(1, 0), (16, 0), (64, 0), (16, 1)])
def test_showcase_all_material_branches(P, O, blue_noise, fetch_min, wf_sort):
    """(wf_sort = 1: the shade stage bins its paths by {finished, regenerating, miss, mesh x specular flag x roulette} in LDS
    before shading them -- active-path sorting; which thread shades a path changes nothing in it)"""
    s = P.Scene(96, 64)
    P.scenes.showcase(s, segments=12)
    s.set_option("wavefront", 1)
    s.set_option("wf_sort", wf_sort)
    s.set_option("fetch_min", fetch_min)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 5, 2)
    assert mode(P, s) == 1
    assert_frames_equal(gpu, cpu)
    s.close()


def test_instanced_meshes_and_thin_lens(P, O, blue_noise):
    """has_transform instances (local-space rays, t rescaling) and a lens that draws random numbers in
    the regenerate step."""
    s = P.Scene(72, 64)
    P.scenes.cornell(s)
    extra = s.addCube(P.Material((0.2, 0.3, 0.9), 0.4))
    s.setPosition(extra, (1.0, -1.0, -5.0))
    s.setRotation(extra, (0.3, 0.5, 0.1))
    s.setInstanceScale(extra, (1.5, 0.7, 1.2))
    ball = s.addSphere(6, P.Material((0.9, 0.9, 0.2), 0.05, 1.0))
    s.setPosition(ball, (-2.0, 1.5, -4.0))
    s.setBVHLeafTarget(2, 0)
    s.set_option("wavefront", 1)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 2)
    if mode(P, s) == 1:  # a small leaf target can make the TLAS a real tree, which the stages do not take
        assert_frames_equal(gpu, cpu)
    s.close()
    s = P.Scene(64, 48)
    P.scenes.cornell(s)
    s.setCamera((0, 0, 5), (0, 0, -5), (0, 1, 0), 40.0, 0.1, 10.0)
    s.set_option("wavefront", 1)
    gpu, cpu = render_both(P, O, s, blue_noise, 3, 4, 2)
    assert mode(P, s) == 1
    assert_frames_equal(gpu, cpu)
    s.close()


def test_band_context(P, O, blue_noise):
    s = P.Scene(64, 64, tile_y0=24, tile_rows=20)
    P.scenes.showcase(s, segments=8)
    s.set_option("wavefront", 1)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 1)
    assert mode(P, s) == 1
    assert_frames_equal(gpu, cpu)
    s.close()


def test_equals_megakernel_at_scale(P, blue_noise):
    """640x360 showcase, 4 spp, 4 bounces, 2 frames: every buffer equals the megakernel's."""
    out = []
    for wf in (0, 1, 2):  # megakernel, wavefront stages, wavefront stages with active-path sorting
        s = P.Scene(640, 360)
        P.scenes.showcase(s)
        s.setPerfSamplesPerPixel(4)
        s.setMaxBounceDepth(4)
        s.setDenoiserEnabled(False)
        s.setBloomEnabled(False)
        s.initBlueNoise()
        s.uploadToGPU()
        s.set_option("count_rays", 1)
        s.set_option("wavefront", 1 if wf else 0)
        s.set_option("wf_sort", 1 if wf == 2 else 0)
        fr = []
        for _ in range(2):
            rgb = s.render_to_host()
            fr.append(dict(accum=s.read(P.BUF_ACCUM), normal=s.read(P.BUF_NORMAL), depth=s.read(P.BUF_DEPTH),
                           object_id=s.read(P.BUF_OBJECT_ID), rgb8=rgb, rng=s.read(P.BUF_RNG), stats=s.stats()))
        assert mode(P, s) == (1 if wf else 0)
        out.append(fr)
        s.close()
    for other in out[1:]:
        for a, b in zip(out[0], other):
            for k in ("accum", "normal", "depth", "object_id", "rgb8", "rng"):
                assert np.array_equal(bits(a[k]), bits(b[k])), k
            assert a["stats"] == b["stats"]

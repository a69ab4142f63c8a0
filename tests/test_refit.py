"""Dynamic geometry (BASELINE config 5): host refit invariants on CPU, GPU refit parity on the box."""
import ctypes as C

import numpy as np
import pytest

from common import assert_frames_equal, render_both


def node_boxes(P, d, m):
    M = d.contents.meshes[m]
    return np.ctypeslib.as_array(C.cast(M.nodes, C.POINTER(C.c_float)), (M.node_count, 10))[:, :6].copy()


def test_host_refit_of_unchanged_vertices_is_the_built_tree(P):
    s = P.Scene(32, 32, device=P.HOST_ONLY)
    w, ship = P.scenes.fluid(s, cells=24, t=0.0, ship_segments=10)
    d = s.flatten()
    before = [node_boxes(P, d, m) for m in (w, ship)]
    tl = d.contents.tlas_nodes[0]
    t_before = (tl.bmin.x, tl.bmin.y, tl.bmin.z, tl.bmax.x, tl.bmax.y, tl.bmax.z)
    # moving the vertices and moving them back must restore every box bit for bit
    v0 = P.scenes.water_vertices(24, 0.0)
    s.setVertices(w, P.scenes.water_vertices(24, 0.7))
    with pytest.raises(P.PtrtError):
        s.refitObjectChanges()          # host-only scene: no back end, fails loudly
    s.setVertices(w, v0)
    d2 = s.flatten()                    # (rebuild path: same vertices -> same tree)
    for m, b in zip((w, ship), before):
        assert np.array_equal(node_boxes(P, d2, m).view(np.uint32), b.view(np.uint32))
    tl = d2.contents.tlas_nodes[0]
    assert (tl.bmin.x, tl.bmin.y, tl.bmin.z, tl.bmax.x, tl.bmax.y, tl.bmax.z) == t_before


@pytest.mark.gpu
def test_gpu_refit_matches_oracle_on_host_refit(P, O, blue_noise):
    s = P.Scene(96, 64)
    w, ship = P.scenes.fluid(s, cells=24, t=0.0, ship_segments=10)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 1)
    assert_frames_equal(gpu, cpu)
    base_ids = gpu[0]["object_id"].copy()
    for step, t in enumerate((0.4, 1.1)):
        s.setVertices(w, P.scenes.water_vertices(24, t))
        s.refitObjectChanges()
        assert s.getFrameCount() == 0
        s.reset_rng(P.DEFAULT_SEED)
        rng = O.xorwow_init(P.DEFAULT_SEED, 0, 96 * 64)
        rgb = s.render_to_host()
        c = O.render(s.flatten(), 96, 64, 2, 4, 0, blue_noise, rng, threads=8)
        assert np.array_equal(s.read(P.BUF_OBJECT_ID), c["object_id"])
        assert np.array_equal(s.read(P.BUF_ACCUM).view(np.uint32), c["accum"].view(np.uint32))
        assert np.array_equal(s.read(P.BUF_DEPTH).view(np.uint32), c["depth"].view(np.uint32))
        assert np.array_equal(s.read(P.BUF_RNG), rng)
        assert np.array_equal(rgb, O.tonemap(c["accum"], 96, 64))
        assert not np.array_equal(s.read(P.BUF_DEPTH), gpu[0]["depth"])      # the surface did move
    # refit vs the reference's behaviour (full rebuild of the moved mesh): same first hits but for ties
    refit_ids, refit_depth = s.read(P.BUF_OBJECT_ID), s.read(P.BUF_DEPTH)
    r = P.Scene(96, 64)
    P.scenes.fluid(r, cells=24, t=1.1, ship_segments=10)
    g2, _ = render_both(P, O, r, blue_noise, 2, 4, 1)
    assert (g2[0]["object_id"] == refit_ids).mean() > 0.999
    assert np.allclose(g2[0]["depth"], refit_depth, rtol=1e-5, atol=1e-5) or \
        (np.abs(g2[0]["depth"] - refit_depth) < 1e-4).mean() > 0.999
    s.close()
    r.close()


@pytest.mark.gpu
def test_refit_from_device_memory(P, O, blue_noise):
    import torch
    s = P.Scene(64, 48)
    w, ship = P.scenes.fluid(s, cells=16, t=0.0, ship_segments=8)
    gpu, cpu = render_both(P, O, s, blue_noise, 1, 3, 1)
    assert_frames_equal(gpu, cpu)
    v = P.scenes.water_vertices(16, 0.9)
    s.setVertices(w, v)
    s.refitObjectChanges()
    s.reset_rng(P.DEFAULT_SEED)
    a = s.render_to_host().copy()
    acc_a = s.read(P.BUF_ACCUM).copy()
    # same positions, this time handed over as a device buffer after wrecking them on the GPU first
    junk = torch.from_numpy(np.ascontiguousarray(P.scenes.water_vertices(16, 3.3))).cuda()
    s.refitFromDevice(w, junk.data_ptr())
    good = torch.from_numpy(np.ascontiguousarray(v)).cuda()
    s.refitFromDevice(w, good.data_ptr())
    s.reset_rng(P.DEFAULT_SEED)
    b = s.render_to_host()
    assert np.array_equal(a, b) and np.array_equal(acc_a.view(np.uint32), s.read(P.BUF_ACCUM).view(np.uint32))
    # wrong vertex count is rejected
    assert P.lib.ptrt_update_vertices(s.ctx, w, good.data_ptr() and None, 7, 1) == -1
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [False, True])
def test_refit_from_host_memory_does_not_wait_and_does_not_keep_the_buffer(P, O, blue_noise, pinned):
    """ptrt_update_vertices from host memory returns without waiting for the stream (ordinary memory: staged in pinned memory
    of the context; pinned memory: a copy stream + device staging) and the caller's buffer is its own again at once: it is
    overwritten with other positions right after every call -- five updates in a row, more than the staging ring holds, with
    frames in flight -- and the frames still are the ones of the positions handed over."""
    import torch
    W, H = 96, 64
    times = [0.2, 0.9, 1.7, 2.4, 3.1]
    want = []
    ref = P.Scene(W, H)
    w, _ = P.scenes.fluid(ref, cells=24, t=0.0, ship_segments=8)
    render_both(P, O, ref, blue_noise, 1, 3, 1)
    for t in times:
        ref.setVertices(w, P.scenes.water_vertices(24, t))
        ref.refitObjectChanges()
        want.append(ref.render_to_host().copy())
    ref.close()
    s = P.Scene(W, H)
    w, _ = P.scenes.fluid(s, cells=24, t=0.0, ship_segments=8)
    render_both(P, O, s, blue_noise, 1, 3, 1)
    n = P.scenes.water_vertices(24, 0.0).size
    buf = torch.empty(n, dtype=torch.float32)
    if pinned:
        buf = buf.pin_memory()
    dev = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in times]
    for k, t in enumerate(times):
        buf.numpy()[:] = np.ascontiguousarray(P.scenes.water_vertices(24, t)).ravel()
        s.refitFromHost(w, buf.data_ptr())
        buf.numpy()[:] = 1e9  # the caller's buffer is the caller's again
        s.render_to_device(dev[k].data_ptr())
    s.sync()
    for k in range(len(times)):
        assert np.array_equal(dev[k].cpu().numpy(), want[k]), f"frame {k}"
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("rebuild", [False, True])
def test_refit_and_rebuild_behind_a_real_tlas(P, O, blue_noise, rebuild):
    """A vertex-animated mesh among > 17 meshes: BLAS refit (or GPU rebuild) on the GPU, TLAS rebuilt by the host
    over the new boxes and handed over with ptrt_update_instances; frames equal the oracle's over the host copy."""
    from test_parity_gpu import _many_meshes
    from common import assert_frames_equal
    W, H, spp, depth = 80, 60, 2, 4
    s = P.Scene(W, H)
    _many_meshes(P, s, n=24)
    blob = s.addSphere(16, P.Material((0.8, 0.4, 0.1), 0.3))
    s.setPerfSamplesPerPixel(spp)
    s.setMaxBounceDepth(depth)
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    s.initBlueNoise()
    s.uploadToGPU()
    s.reset_rng(P.DEFAULT_SEED)
    s.set_option("count_rays", 1)
    import ctypes
    md = ctypes.cast(s.flatten(), ctypes.POINTER(P.SceneDesc)).contents.meshes[blob]
    base = np.ctypeslib.as_array(ctypes.cast(md.verts, ctypes.POINTER(ctypes.c_float)), (md.vert_count, 3)).copy()
    rng = O.xorwow_init(P.DEFAULT_SEED, 0, W * H)
    for f in range(3):
        if f:
            v = base * np.float32(1.0 + 0.4 * f) + np.array([0.6 * f, -0.3 * f, -4.0 - f], np.float32)
            s.setVertices(blob, v.astype(np.float32))
            (s.rebuildObjectChanges if rebuild else s.refitObjectChanges)()
        fc = s.getFrameCount()
        rgb = s.render_to_host()
        g = dict(accum=s.read(P.BUF_ACCUM), normal=s.read(P.BUF_NORMAL), depth=s.read(P.BUF_DEPTH),
                 object_id=s.read(P.BUF_OBJECT_ID), rgb8=rgb, rng=s.read(P.BUF_RNG), stats=s.stats())
        c = O.render(s.flatten(), W, H, spp, depth, fc, blue_noise, rng, threads=8)
        c["rgb8"] = O.tonemap(c["accum"], W, H, threads=8)
        c["rng"] = rng.copy()
        assert_frames_equal([g], [c])
    assert (g["object_id"] == blob).sum() > 20
    s.close()


@pytest.mark.gpu
def test_fluid_full_size_refit_frame_equals_the_oracle(P, O, blue_noise):
    """BASELINE configs[4] at its full size (131,072-triangle water surface, 1920x1080, 2 spp): the built frame, then the
    frame after a vertex update + GPU refit, against the oracle on the host-refitted tree -- every buffer and generator state."""
    W, H = 1920, 1080
    s = P.Scene(W, H)
    w, _ = P.scenes.fluid(s, cells=256, t=0.0)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 1, threads=16)
    assert_frames_equal(gpu, cpu)
    s.setVertices(w, P.scenes.water_vertices(256, 0.7))
    s.refitObjectChanges()
    s.reset_rng(P.DEFAULT_SEED)
    rng = O.xorwow_init(P.DEFAULT_SEED, 0, W * H)
    rgb = s.render_to_host()
    c = O.render(s.flatten(), W, H, 2, 4, 0, blue_noise, rng, threads=16)
    assert np.array_equal(s.read(P.BUF_OBJECT_ID), c["object_id"])
    for k, b in (("accum", P.BUF_ACCUM), ("depth", P.BUF_DEPTH), ("normal", P.BUF_NORMAL)):
        assert np.array_equal(s.read(b).view(np.uint32), c[k].view(np.uint32)), k
    assert np.array_equal(s.read(P.BUF_RNG), rng)
    assert np.array_equal(rgb, O.tonemap(c["accum"], W, H, threads=16))
    assert not np.array_equal(s.read(P.BUF_DEPTH), gpu[0]["depth"])  # the surface did move
    s.close()

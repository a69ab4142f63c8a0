"""GPU BVH rebuild (ptrt_build_bvh / ptrt_update_triangles).

The rebuilt tree is read back (prim order) and the oracle traverses the SAME tree, so frames are
compared bit for bit; the sort itself is checked against a numpy restatement of the Morton keys.
PARITY UNPINNED w.r.t. the CUDA reference (it has no GPU builder; its CPU builder's leaf assignment
depends on std::nth_element)."""
import numpy as np
import pytest

from common import assert_frames_equal, render_both

pytestmark = pytest.mark.gpu


def morton_order(verts, faces):
    """The face order ptrt_build_bvh must produce: stable sort by 30-bit Morton code of the fp32 centroid."""
    v = verts.astype(np.float32)
    c = ((v[faces[:, 0]] + v[faces[:, 1]]) + v[faces[:, 2]]) * np.float32(1.0 / 3.0)
    lo, hi = c.min(axis=0), c.max(axis=0)
    ext = np.float32((hi - lo).astype(np.float32).max())         # one scale for all axes: the largest extent
    q = np.zeros(c.shape, dtype=np.int64)
    for k in range(3):
        if ext > 0:
            t = ((c[:, k] - lo[k]) / ext).astype(np.float32)
            t = np.clip(t, np.float32(0), np.float32(1))
            q[:, k] = np.minimum((t * np.float32(1024.0)).astype(np.int64), 1023)

    def spread(x):
        out = np.zeros_like(x)
        for b in range(10):
            out |= ((x >> b) & 1) << (3 * b)
        return out
    keys = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
    return np.argsort(keys, kind="stable").astype(np.int32), keys


def mesh_arrays(P, s, m):
    import ctypes as C
    M = s.flatten().contents.meshes[m]
    verts = np.ctypeslib.as_array(C.cast(M.verts, C.POINTER(C.c_float)), (M.vert_count, 3)).copy()
    faces = np.ctypeslib.as_array(C.cast(M.faces, C.POINTER(C.c_int32)), (M.face_count, 3)).copy()
    return verts, faces


def render_pair(P, O, s, blue_noise, spp=2, depth=4):
    W, H = s.width, s.height
    s.reset_rng(P.DEFAULT_SEED)
    rng = O.xorwow_init(P.DEFAULT_SEED, 0, W * H)
    fc = s.getFrameCount()
    rgb = s.render_to_host()
    c = O.render(s.flatten(), W, H, spp, depth, fc, blue_noise, rng, threads=8)
    g = dict(accum=s.read(P.BUF_ACCUM), normal=s.read(P.BUF_NORMAL), depth=s.read(P.BUF_DEPTH),
             object_id=s.read(P.BUF_OBJECT_ID), rgb8=rgb, rng=s.read(P.BUF_RNG))
    c["rgb8"] = O.tonemap(c["accum"], W, H)
    c["rng"] = rng
    return g, c


def test_rebuild_sorts_faces_in_morton_order_and_frames_match_the_oracle(P, O, blue_noise):
    s = P.Scene(96, 64)
    w, ship = P.scenes.fluid(s, cells=40, t=0.0, ship_segments=24)      # 3200-triangle soup + 1152-triangle sphere
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 1)
    assert_frames_equal(gpu, cpu)
    host_order = s.primIndices(w).copy()
    for t, graphs in ((0.6, 1), (1.1, 1), (1.7, 0)):     # captured-graph replay (twice) and plain launches
        s.set_option("use_graphs", graphs)
        s.setVertices(w, P.scenes.water_vertices(40, t))
        s.rebuildObjectChanges()
        verts, faces = mesh_arrays(P, s, w)
        want, keys = morton_order(verts, faces)
        got = s.primIndices(w)
        assert np.array_equal(np.sort(got), np.arange(len(got)))            # a permutation of the faces
        assert np.array_equal(got, want)                                     # ... in stable Morton order
        assert not np.array_equal(got, host_order)
        assert len(np.unique(keys)) > len(keys) // 4                         # the codes do spread
        g, c = render_pair(P, O, s, blue_noise)
        assert_frames_equal([g], [c], check_stats=False)
    # a rebuilt tree finds the same surfaces as the reference's behaviour (fresh host build), ties aside
    r = P.Scene(96, 64)
    P.scenes.fluid(r, cells=40, t=1.7, ship_segments=24)
    g2, _ = render_both(P, O, r, blue_noise, 2, 4, 1)
    same = (g2[0]["depth"].view(np.uint32) == g["depth"].view(np.uint32)) & (g2[0]["object_id"] == g["object_id"])
    assert same.mean() > 0.999
    # the static, indexed sphere mesh can be rebuilt too (shared vertices), and nothing changes but ties
    s.scale(ship, (1.0, 1.0, 1.0))       # marks it dirty without moving it
    s.rebuildObjectChanges()
    verts, faces = mesh_arrays(P, s, ship)
    assert np.array_equal(s.primIndices(ship), morton_order(verts, faces)[0])
    g3, c3 = render_pair(P, O, s, blue_noise)
    assert_frames_equal([g3], [c3], check_stats=False)
    s.close()
    r.close()


def test_radix_sort_at_scale_and_rebuild_from_device(P, O, blue_noise):
    """131,072-triangle water (BASELINE config 5 size): multi-workgroup sort, positions fed from device memory."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")      # the runtime libptrt_amd.so itself is linked against
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    s = P.Scene(128, 72)
    w, ship = P.scenes.fluid(s, cells=256, t=0.0, ship_segments=16)
    s.setPerfSamplesPerPixel(1)
    s.setMaxBounceDepth(3)
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    s.initBlueNoise()
    s.uploadToGPU()
    v = P.scenes.water_vertices(256, 0.9)
    dv = C.c_void_p()
    assert hip.hipMalloc(C.byref(dv), v.nbytes) == 0
    assert hip.hipMemcpy(dv, v.ctypes.data_as(C.c_void_p), v.nbytes, 1) == 0     # hipMemcpyHostToDevice
    s.rebuildFromDevice(w, dv.value)
    s.sync()
    hip.hipFree(dv)
    got = np.zeros(131072, np.int32)
    assert P.lib.ptrt_read_prim_order(s.ctx, w, got.ctypes.data_as(C.POINTER(C.c_int)), got.size) == 0
    faces = np.arange(131072 * 3, dtype=np.int32).reshape(-1, 3)
    want, _ = morton_order(v, faces)
    assert np.array_equal(got, want)
    # host copy follows (vertices + prim order), then the oracle sees the same tree
    s.setVertices(w, v)
    s.rebuildObjectChanges()
    assert np.array_equal(s.primIndices(w), want)
    g, c = render_pair(P, O, s, blue_noise, spp=1, depth=3)
    assert_frames_equal([g], [c], check_stats=False)
    s.close()


def test_update_triangles_with_fewer_triangles(P, O, blue_noise):
    s = P.Scene(96, 64)
    w, ship = P.scenes.fluid(s, cells=32, t=0.0, ship_segments=12)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 1)
    assert_frames_equal(gpu, cpu)
    full = P.scenes.water_vertices(32, 0.5).reshape(-1, 9)
    keep = full[(np.arange(len(full)) % 5) != 0]                    # a surface with holes: 1639 of 2048 triangles
    s.updateTriangles(w, keep)
    verts, faces = mesh_arrays(P, s, w)
    assert np.array_equal(verts[: len(keep) * 3], keep.reshape(-1, 3))
    assert (verts[len(keep) * 3:] == keep.reshape(-1, 3)[-1]).all()  # padding = the last real vertex
    g, c = render_pair(P, O, s, blue_noise)
    assert_frames_equal([g], [c], check_stats=False)
    # same picture as a scene that only ever had the kept triangles (ties aside)
    r = P.Scene(96, 64)
    mat = P.Material((1.0, 1.0, 1.0), 0.0, transmission=1.0, ior=1.33, specular=(0.04, 0.04, 0.04))
    r.addTriangles(keep, mat)
    sh = r.addSphere(12, P.Material((0.6, 0.35, 0.2), 0.5))
    r.scale(sh, (6.0, 2.0, 3.0))
    r.moveTo(sh, (0.0, 0.4, -2.0))
    r.setSkyGradient((0.35, 0.55, 0.95), (0.9, 0.95, 1.0))
    r.addDirectionalLight((-0.4, -1.0, -0.3), (1.0, 0.96, 0.9), 3.0)
    r.setCamera((0.0, 6.0, 18.0), (0.0, 0.0, 0.0), (0, 1, 0), 45.0)
    g2, _ = render_both(P, O, r, blue_noise, 2, 4, 1)
    same = (g2[0]["depth"].view(np.uint32) == g["depth"].view(np.uint32)) & (g2[0]["object_id"] == g["object_id"])
    assert same.mean() > 0.999
    # back to the full count, and the error paths
    s.updateTriangles(w, full)
    g, c = render_pair(P, O, s, blue_noise)
    assert_frames_equal([g], [c], check_stats=False)
    with pytest.raises(P.PtrtError, match="room for"):
        s.updateTriangles(w, np.concatenate([full, full[:1]]))
    with pytest.raises(P.PtrtError, match="soup"):
        s.updateTriangles(ship, full[:4])
    assert P.lib.ptrt_build_bvh(s.ctx, 99) == -1
    s.close()
    r.close()

"""Moving instances (Mesh::setPosition / setRotation + Scene::commitObjectChanges): the transform-dirty case of
updateAccelerationStructures goes through ptrt_update_instances -- new matrices and TLAS, triangles untouched -- and
every frame still equals the oracle's render of the same host structures, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from common import assert_frames_equal

pytestmark = pytest.mark.gpu


def _counts(P, s):
    out = (C.c_int * 2)()
    P.lib.ptrt_debug_upload_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    assert P.lib.ptrt_debug_upload_counts(s.ctx, out) == 0
    return out[0], out[1]


@pytest.mark.parametrize("many", [False, True])
def test_moving_instances_take_the_fast_path_and_match_the_oracle(P, O, blue_noise, many):
    W, H, spp, depth = 88, 64, 2, 4
    s = P.Scene(W, H)
    if many:  # > 17 meshes: a real TLAS whose topology changes as the instances move
        from test_parity_gpu import _many_meshes
        _many_meshes(P, s, n=30)
    else:
        P.scenes.cornell(s)
    cube = s.addCube(P.Material((0.2, 0.3, 0.9), 0.4))
    s.setPosition(cube, (1.0, -1.0, -5.0))
    s.setRotation(cube, (0.3, 0.5, 0.1))
    s.setInstanceScale(cube, (1.2, 0.7, 1.0))
    ball = s.addSphere(8, P.Material((0.9, 0.9, 0.2), 0.05, 1.0))
    s.setPosition(ball, (-2.0, 1.5, -4.0))
    s.setPerfSamplesPerPixel(spp)
    s.setMaxBounceDepth(depth)
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    s.initBlueNoise()
    s.uploadToGPU()
    s.reset_rng(P.DEFAULT_SEED)
    s.set_option("count_rays", 1)
    rng = O.xorwow_init(P.DEFAULT_SEED, 0, W * H)
    assert _counts(P, s) == (1, 0)
    for f in range(4):
        if f:
            s.setPosition(cube, (1.0 - 0.8 * f, -1.0 + 0.3 * f, -5.0 + 0.5 * f))
            s.setRotation(cube, (0.3 + 0.4 * f, 0.5, 0.1 * f))
            s.setPosition(ball, (-2.0 + 0.9 * f, 1.5 - 0.6 * f, -4.0))
            s.commitObjectChanges()
        fc = s.getFrameCount()
        rgb = s.render_to_host()
        g = dict(accum=s.read(P.BUF_ACCUM), normal=s.read(P.BUF_NORMAL), depth=s.read(P.BUF_DEPTH),
                 object_id=s.read(P.BUF_OBJECT_ID), rgb8=rgb, rng=s.read(P.BUF_RNG), stats=s.stats())
        c = O.render(s.flatten(), W, H, spp, depth, fc, blue_noise, rng, threads=8)
        c["rgb8"] = O.tonemap(c["accum"], W, H, threads=8)
        c["rng"] = rng.copy()
        assert_frames_equal([g], [c])
        assert (g["object_id"] == cube).sum() > 20, f
    assert _counts(P, s) == (1, 3)      # one full upload, three transform-only updates
    # a different mesh count is refused (that is a geometry change) and nothing is uploaded
    d = C.cast(s.flatten(), C.POINTER(P.SceneDesc)).contents
    P.lib.ptrt_update_instances.argtypes = [C.c_void_p, C.POINTER(P.MeshDesc), C.c_int, C.POINTER(P.BvhNode), C.c_int,
                                            C.POINTER(C.c_int32), C.c_int]
    assert P.lib.ptrt_update_instances(s.ctx, d.meshes, d.mesh_count - 1, d.tlas_nodes, d.tlas_node_count,
                                       d.tlas_mesh_indices, d.tlas_index_count) == -1
    assert b"use ptrt_upload_geometry" in P.lib.ptrt_last_error(s.ctx)
    assert _counts(P, s) == (1, 3)
    # a vertex change still takes the full path
    s.scale(cube, (1.1, 1.0, 1.0))
    s.commitObjectChanges()
    assert _counts(P, s) == (2, 3)
    s.close()

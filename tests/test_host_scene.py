"""Host-side mirror of the reference's Scene/Mesh API (host/ptrt/*.hpp): geometry generators,
vertex-baking transforms, BVH builder invariants, flattening, blue-noise table."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def np_mesh(d, m):
    M = d.contents.meshes[m]
    v = np.ctypeslib.as_array(C.cast(M.verts, C.POINTER(C.c_float)), (M.vert_count, 3)).copy()
    f = np.ctypeslib.as_array(C.cast(M.faces, C.POINTER(C.c_int32)), (M.face_count, 3)).copy()
    n = np.ctypeslib.as_array(C.cast(M.nodes, C.POINTER(C.c_int32)), (M.node_count, 10)).copy()
    p = np.ctypeslib.as_array(M.prim_indices, (M.prim_count,)).copy()
    return v, f, n, p, M


def test_cube_and_transform_order(P):
    """scale -> moveTo -> rotateSelfEulerXYZ bake into the vertices in that order (mesh.cuh:548-640)."""
    s = P.Scene(32, 32, device=P.HOST_ONLY)
    m = s.addCube(P.Material((1, 0, 0)))
    s.scale(m, (1.5, 3.0, 1.5))
    s.moveTo(m, (-1.5, -3.5, -6))
    s.rotateSelfEulerXYZ(m, (0, 0.3, 0))
    v, f, n, p, M = np_mesh(s.flatten(), 0)
    f32 = np.float32
    base = np.array([[-.5, -.5, -.5], [.5, -.5, -.5], [.5, .5, -.5], [-.5, .5, -.5],
                     [-.5, -.5, .5], [.5, -.5, .5], [.5, .5, .5], [-.5, .5, .5]], f32)
    w = base * np.array([1.5, 3.0, 1.5], f32)
    c = (w.min(0) + w.max(0)) * f32(0.5)
    w = w + (np.array([-1.5, -3.5, -6], f32) - c)
    c = (w.min(0) + w.max(0)) * f32(0.5)
    q = w - c
    cy, sy = f32(np.cos(f32(0.3))), f32(np.sin(f32(0.3)))   # glibc cosf/sinf == numpy float32 here
    x2 = cy * q[:, 0] + sy * q[:, 2]
    z2 = -sy * q[:, 0] + cy * q[:, 2]
    expect = np.stack([x2, q[:, 1], z2], 1).astype(f32) + c
    assert np.allclose(v, expect, rtol=0, atol=2e-6)
    assert f.tolist()[:4] == [[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7]] and len(f) == 12
    assert M.has_transform == 0 and n.shape[0] == 1 and n[0, 9] == 12  # one leaf of 12 (<= 17)
    assert sorted(p.tolist()) == list(range(12))


def test_sphere_counts_and_plane(P):
    s = P.Scene(32, 32, device=P.HOST_ONLY)
    a = s.addSphere(71, P.Material((1, 1, 1)))
    b = s.addPlaneXZ(-3.0, 50.0, P.Material((0.8, 0.8, 0.8)))
    assert s.meshCounts(a)[:2] == (72 * 72, 2 * 71 * 71)       # 5,184 vertices / 10,082 triangles
    d = s.flatten()
    v, f, n, p, M = np_mesh(d, b)
    assert v.shape == (6, 3) and np.all(v[:, 1] == -3.0) and f.tolist() == [[0, 1, 2], [3, 4, 5]]
    # CCW from +Y: normal +Y
    nrm = np.cross(v[1] - v[0], v[2] - v[0])
    assert nrm[1] > 0 and nrm[0] == 0 and nrm[2] == 0
    r = np.linalg.norm(np_mesh(d, a)[0], axis=1)
    assert np.allclose(r, 0.5, atol=1e-6)


@pytest.mark.parametrize("leaf", [(12, 5), (4, 0), (1, 0)])
def test_bvh_invariants(P, leaf):
    s = P.Scene(32, 32, device=P.HOST_ONLY)
    P.scenes.showcase(s, segments=10)
    s.setBVHLeafTarget(*leaf)
    d = s.flatten()
    leaf_max = leaf[0] + leaf[1]
    for m in range(d.contents.mesh_count):
        v, f, nodes, prims, M = np_mesh(d, m)
        assert sorted(prims.tolist()) == list(range(len(f)))            # a permutation of the faces
        boxes = np.ctypeslib.as_array(C.cast(M.nodes, C.POINTER(C.c_float)), (M.node_count, 10))[:, :6]
        seen = np.zeros(len(nodes), bool)
        stack = [(0, 0)]
        while stack:
            i, depth = stack.pop()
            assert not seen[i]
            seen[i] = True
            left, right, start, count = nodes[i, 6:10]
            if count > 0:
                assert count <= leaf_max and left == -1 and right == -1
                tri = v[f[prims[start:start + count]]].reshape(-1, 3)
                assert np.all(tri >= boxes[i, :3]) and np.all(tri <= boxes[i, 3:])
            else:
                assert left == i + 1 and right > left                       # pre-order
                for ch in (left, right):
                    assert np.all(boxes[ch, :3] >= boxes[i, :3]) and np.all(boxes[ch, 3:] <= boxes[i, 3:])
                    stack.append((ch, depth + 1))
            assert depth <= 23
        assert seen.all()
    # TLAS covers every mesh exactly once
    ids = np.ctypeslib.as_array(d.contents.tlas_mesh_indices, (d.contents.tlas_index_count,))
    assert sorted(ids.tolist()) == list(range(d.contents.mesh_count))


def test_has_transform_flag_and_matrices(P):
    s = P.Scene(32, 32, device=P.HOST_ONLY)
    a = s.addCube(P.Material((1, 1, 1)))
    b = s.addCube(P.Material((1, 1, 1)))
    c = s.addCube(P.Material((1, 1, 1)))
    s.setPosition(b, (2.0, 1.0, -3.0))
    s.setPosition(c, (2.0, 0.0, -3.0))
    s.setRotation(c, (0.0, 0.5, 0.0))
    d = s.flatten()
    Ma, Mb, Mc = d.contents.meshes[a], d.contents.meshes[b], d.contents.meshes[c]
    assert Ma.has_transform == 0 and Mb.has_transform == 1 and Mc.has_transform == 1
    mat = lambda m: np.array(list(m), np.float32).reshape(4, 4)
    w, inv = mat(Mb.world), mat(Mb.inverse)
    assert np.allclose(w[:3, 3], [2, 1, -3])                              # row-major translation column
    assert np.allclose(w @ inv, np.eye(4), atol=1e-6)                     # pure translation: a true inverse
    assert np.array_equal(mat(Mb.normal), inv.T)
    # Rotation + x-translation: the reference's mat4::inverse uses A0113/A0112 where the cofactor
    # expansion needs A0213/A0212 (mat4.cuh:245-246), so entries m[6], m[7] are NOT those of the
    # true inverse whenever m[3] != 0.  The mirror reproduces the formula as written.
    w, inv = mat(Mc.world), mat(Mc.inverse)
    true_inv = np.linalg.inv(w.astype(np.float64))
    diff = np.abs(inv - true_inv) > 1e-5
    assert diff.reshape(-1).nonzero()[0].tolist() == [7] and abs(inv[1, 3] - 2.0 * np.sin(np.float32(0.5))) < 1e-6
    # the TLAS root bounds the moved instances
    t = d.contents.tlas_nodes[0]
    assert t.bmax.x >= 2.0 and t.bmin.z <= -3.0


def test_lights_materials_camera_flatten(P):
    s = P.Scene(192, 108, device=P.HOST_ONLY)
    P.scenes.showcase(s, segments=6)
    d = s.flatten().contents
    assert d.mesh_count == 11 and d.light_count == 6 and d.materials.count == 11 and d.use_sky == 1
    L = d.lights[0]
    assert L.type == 2 and abs(L.inner_cone - np.cos(np.float32(0.1))) < 1e-7 and abs(L.outer_cone - np.cos(np.float32(0.8))) < 1e-7
    assert abs(L.direction.y + 1.0) < 1e-7 and L.radius == np.float32(0.1)
    assert d.materials.transmission[3] == np.float32(0.95) and d.materials.iridescence_thickness[3] == 400.0
    assert d.materials.metallic[5] == np.float32(0.4) and abs(d.materials.specular[5].x - 0.04) < 1e-7
    cam = d.camera
    assert cam.lens_radius == 0.0 and (cam.origin.x, cam.origin.y, cam.origin.z) == (0.0, 2.0, 5.0)
    # horizontal/vertical ratio is the aspect
    hl = np.sqrt(cam.horizontal.x ** 2 + cam.horizontal.y ** 2 + cam.horizontal.z ** 2)
    vl = np.sqrt(cam.vertical.x ** 2 + cam.vertical.y ** 2 + cam.vertical.z ** 2)
    assert abs(hl / vl - 192 / 108) < 1e-5


def test_presets_and_setters(P):
    s = P.Scene(64, 64, device=P.HOST_ONLY)
    assert s.settings() == dict(spp=1, depth=4, denoiser=True, bloom=True, scale=1.0)
    s.setPerformancePreset("ultra")
    assert s.settings()["spp"] == 128 and s.settings()["depth"] == 32 and not s.settings()["denoiser"]
    s.setPerformancePreset("fast")
    st = s.settings()
    assert st["depth"] == 2 and abs(st["scale"] - 0.35) < 1e-7 and st["spp"] == 128  # presets other than ultra keep spp
    s.setMaxBounceDepth(99)
    assert s.settings()["depth"] == 16
    s.setMaxBounceDepth(0)
    assert s.settings()["depth"] == 1


def test_blue_noise_table_is_pinned(P, blue_noise):
    meta = json.load(open(os.path.join(GOLD, "blue_noise_libstdcxx.json")))
    assert blue_noise.shape == (8192,) and blue_noise.min() >= 0.0 and blue_noise.max() < 1.0
    assert hashlib.sha256(blue_noise.tobytes()).hexdigest() == meta["sha256"]
    assert np.array_equal(blue_noise[:64], np.array(meta["first64"], np.float32))
    # stratification survives the relaxation: point (x,y) stays near cell (x,y)
    pts = blue_noise.reshape(64, 64, 2)
    gx, gy = np.meshgrid(np.arange(64), np.arange(64))
    dx, dy = np.abs(pts[..., 0] * 64 - (gx + 0.5)), np.abs(pts[..., 1] * 64 - (gy + 0.5))
    assert np.all(np.minimum(dx, 64 - dx) < 1.0) and np.all(np.minimum(dy, 64 - dy) < 1.0)  # toroidal domain


def test_obj_loader(P, tmp_path):
    obj = tmp_path / "quad.obj"
    obj.write_text("# quad\nv 0 0 0\nv 2 0 0\nv 2 2 0\nv 0 2 0\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4 -3 -2\n")
    s = P.Scene(16, 16, device=P.HOST_ONLY)
    m = s.addMesh(str(obj), P.Material((1, 1, 1)))
    v, f, n, p, M = np_mesh(s.flatten(), m)
    assert f.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2]]           # fan + negative indices
    assert np.allclose(v.mean(0), 0, atol=1e-7) and v.min() == -1.0  # re-centred on the vertex mean
    with pytest.raises(P.PtrtError):
        s.addMesh(str(tmp_path / "missing.obj"), P.Material((1, 1, 1)))
    # every corner form, CRLF line ends, records to ignore, a short `v` line, exponents, no newline at the end
    obj2 = tmp_path / "forms.obj"
    obj2.write_bytes(b"o thing\r\nv 1e0 0 0\r\nv 3 0 0.0\r\nv 3 2 0\r\nv 1 2 -0\r\nv 9 9\r\nvt 0.5 0.5\r\nvn 0 0 1\r\nusemtl m\r\n"
                     b"f 1//1 2//1 3//1\r\nf 1/2 3/4 4/1\r\n\tf\t-1 -4 -3\r\nf 1 2\r\nf 4 3 2 1")
    m2 = s.addMesh(str(obj2), P.Material((1, 1, 1)))
    v2, f2, *_ = np_mesh(s.flatten(), m2)
    assert v2.shape == (4, 3) and np.allclose(v2.mean(0), 0, atol=1e-7) and np.allclose(v2[0], (-1, -1, 0))
    assert f2.tolist() == [[0, 1, 2], [0, 2, 3], [3, 0, 1], [3, 2, 1], [3, 1, 0]]
    # what `istream >> float / int` refuses is refused: nan, inf, hexadecimal and overflowing reals drop their `v` record
    # (the later vertices keep their numbers), an index outside int ends its `f` record; ".5", "5.", "+1", "1E+0" are reals
    obj3 = tmp_path / "odd.obj"
    obj3.write_text("v nan 0 0\nv 0 inf 0\nv 0x10 0 0\nv 1e99 0 0\nv infinity 1 1\nv .5 5. +1\nv 1E+0 -2e-1 3.25e0\nv 0 0 0\nv 1 1 1\n"
                    "f 1 2 3 99999999999 4\nf 1 2 99999999999999999999999 3 4\nf 4 3 2\n")
    m3 = s.addMesh(str(obj3), P.Material((1, 1, 1)))
    v3, f3, *_ = np_mesh(s.flatten(), m3)
    assert v3.shape == (4, 3)
    assert np.allclose(v3 + np.array([0.625, 1.45, 1.3125], np.float32), [(.5, 5, 1), (1, -.2, 3.25), (0, 0, 0), (1, 1, 1)], atol=1e-6)
    assert f3.tolist() == [[0, 1, 2], [3, 2, 1]]
    empty = tmp_path / "empty.obj"
    empty.write_text("# nothing\nvn 0 0 1\n")
    with pytest.raises(P.PtrtError, match="no valid geometry"):
        s.addMesh(str(empty), P.Material((1, 1, 1)))

"""ptrt_amd -- Python (ctypes) binding of libptrt_amd.so.

Two layers, both plain C underneath:
  * the C ABI of include/ptrt.h (``lib.ptrt_*``): the drop-in boundary of the
    MI355X path-tracing back end;
  * ``Scene``: the reference's host API (``class Scene`` of
    src/pathtracer/scene/scene.cuh:78-2001) as implemented by the C++ mirror in
    host/ptrt/scene.hpp, reached through its flat ``hs_*`` entry points.

Nothing here computes pixels: if the shared library is missing the import fails,
and if there is no HIP device ``Scene(..., device>=0)`` raises.  A ``device=-1``
scene is host-only (build / flatten / inspect) and cannot render.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTRT_AMD_LIB") or os.path.join(_HERE, "libptrt_amd.so")  # env override: A/B builds
if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C ptrt-game-engine_amd` "
        "(or __graft_entry__.build()); there is no fallback implementation"
    )
lib = C.CDLL(LIB_PATH)


def library_info():
    """Path and SHA-256 (first 16 hex digits) of the shared library this process loaded -- PTRT_AMD_LIB can swap it, so a
    measurement names it."""
    import hashlib
    with open(LIB_PATH, "rb") as f:
        return {"path": os.path.relpath(LIB_PATH, os.path.dirname(os.path.dirname(_HERE))), "sha16": hashlib.sha256(f.read()).hexdigest()[:16],
                "from_env": bool(os.environ.get("PTRT_AMD_LIB"))}


PTRT_OK = 0
BUF_ACCUM, BUF_NORMAL, BUF_DEPTH, BUF_OBJECT_ID, BUF_RGB8, BUF_RNG, BUF_DENOISED, BUF_MOTION, BUF_RENDER_ACCUM = range(9)
DEFAULT_SEED = 12345
HOST_ONLY = -1


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class BvhNode(C.Structure):
    _fields_ = [("bmin", Vec3), ("bmax", Vec3), ("left", C.c_int32), ("right", C.c_int32),
                ("start", C.c_int32), ("count", C.c_int32)]


class Tri(C.Structure):
    _fields_ = [("v0", C.c_int32), ("v1", C.c_int32), ("v2", C.c_int32)]


class MeshDesc(C.Structure):
    _fields_ = [("verts", C.POINTER(Vec3)), ("vert_count", C.c_int32),
                ("faces", C.POINTER(Tri)), ("face_count", C.c_int32),
                ("nodes", C.POINTER(BvhNode)), ("node_count", C.c_int32),
                ("prim_indices", C.POINTER(C.c_int32)), ("prim_count", C.c_int32),
                ("world", C.c_float * 16), ("inverse", C.c_float * 16), ("normal", C.c_float * 16),
                ("has_transform", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("position", Vec3), ("direction", Vec3), ("color", Vec3),
                ("intensity", C.c_float), ("range", C.c_float), ("inner_cone", C.c_float),
                ("outer_cone", C.c_float), ("radius", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("origin", Vec3), ("lower_left_corner", Vec3), ("horizontal", Vec3), ("vertical", Vec3),
                ("u", Vec3), ("v", Vec3), ("w", Vec3), ("lens_radius", C.c_float)]


_MAT_FIELDS = ["albedo", "specular", "metallic", "roughness", "emission", "ior", "transmission",
               "transmission_roughness", "clearcoat", "clearcoat_roughness", "subsurface_color",
               "subsurface_radius", "anisotropy", "sheen", "sheen_tint", "iridescence",
               "iridescence_thickness"]
_MAT_VEC = {"albedo", "specular", "emission", "subsurface_color", "sheen_tint"}


class Materials(C.Structure):
    _fields_ = [(n, C.POINTER(Vec3) if n in _MAT_VEC else C.POINTER(C.c_float)) for n in _MAT_FIELDS] + \
               [("count", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(MeshDesc)), ("mesh_count", C.c_int32),
                ("tlas_nodes", C.POINTER(BvhNode)), ("tlas_node_count", C.c_int32),
                ("tlas_mesh_indices", C.POINTER(C.c_int32)), ("tlas_index_count", C.c_int32),
                ("materials", Materials),
                ("lights", C.POINTER(Light)), ("light_count", C.c_int32),
                ("camera", Camera), ("sky_top", Vec3), ("sky_bottom", Vec3), ("use_sky", C.c_int32),
                ("env_rgba", C.POINTER(C.c_float)), ("env_width", C.c_int32), ("env_height", C.c_int32)]


class Hit(C.Structure):
    _fields_ = [("hit", C.c_int32), ("t", C.c_float), ("point", Vec3), ("normal", Vec3),
                ("mesh_index", C.c_int32), ("front_face", C.c_int32), ("u", C.c_float), ("v", C.c_float),
                ("face_index", C.c_int32), ("local_point", Vec3)]


class Stats(C.Structure):
    _fields_ = [("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("paths", C.c_uint64),
                ("shadow_rays_walked", C.c_uint64)]


HIT_DTYPE = np.dtype([("hit", "<i4"), ("t", "<f4"), ("point", "<f4", 3), ("normal", "<f4", 3),
                      ("mesh_index", "<i4"), ("front_face", "<i4"), ("u", "<f4"), ("v", "<f4"),
                      ("face_index", "<i4"), ("local_point", "<f4", 3)])
assert HIT_DTYPE.itemsize == C.sizeof(Hit) == 64

_vp = C.c_void_p
_fp = C.POINTER(C.c_float)


def _sig(name, res, *args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = list(args)
    return f


# ---- include/ptrt.h ------------------------------------------------------------------
_sig("ptrt_abi_version", C.c_int)
_sig("ptrt_last_error", C.c_char_p, _vp)
_sig("ptrt_create", C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp))
_sig("ptrt_destroy", None, _vp)
_sig("ptrt_set_blue_noise", C.c_int, _vp, _fp)
_sig("ptrt_reset_rng", C.c_int, _vp, C.c_ulonglong)
_sig("ptrt_upload_geometry", C.c_int, _vp, C.POINTER(MeshDesc), C.c_int, C.POINTER(BvhNode), C.c_int,
     C.POINTER(C.c_int32), C.c_int)
_sig("ptrt_upload_materials", C.c_int, _vp, C.POINTER(Materials))
_sig("ptrt_upload_lights", C.c_int, _vp, C.POINTER(Light), C.c_int)
_sig("ptrt_set_camera", C.c_int, _vp, C.POINTER(Camera))
_sig("ptrt_set_sky", C.c_int, _vp, C.POINTER(Vec3), C.POINTER(Vec3), C.c_int)
_sig("ptrt_upload_scene", C.c_int, _vp, C.POINTER(SceneDesc))
_sig("ptrt_render", C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int)
_sig("ptrt_sync", C.c_int, _vp)
_sig("ptrt_read_buffer", C.c_int, _vp, C.c_int, _vp, C.c_size_t)
_sig("ptrt_device_buffer", _vp, _vp, C.c_int)
_sig("ptrt_write_rng", C.c_int, _vp, C.POINTER(C.c_uint32), C.c_size_t)
_sig("ptrt_trace_rays", C.c_int, _vp, _fp, _fp, C.c_int, _vp)
_sig("ptrt_get_stats", C.c_int, _vp, C.POINTER(Stats))
_sig("ptrt_set_option", C.c_int, _vp, C.c_char_p, C.c_longlong)
_sig("ptrt_get_option", C.c_int, _vp, C.c_char_p, C.POINTER(C.c_longlong))
_sig("ptrt_last_kernel_ms", C.c_int, _vp, _fp, _fp)
_sig("ptrt_set_stream", C.c_int, _vp, _vp)
_sig("ptrt_kernel_ms_history", C.c_int, _vp, _fp, C.c_int)
_sig("ptrt_launch_ms_history", C.c_int, _vp, _fp, _fp, C.c_int)
_sig("ptrt_debug_detmath", C.c_int, _vp, C.c_int, _fp, _fp, C.c_int, _fp)

# ---- Scene mirror (csrc/ptrt_host_capi.cpp) ---------------------------------------------
_sig("hs_last_error", C.c_char_p)
_sig("hs_material_default", None, _fp)
_sig("hs_material_make", None, _fp, C.c_float, C.c_float, _fp)
_sig("hs_scene_create", _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
_sig("hs_scene_create_interleaved", _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
_sig("hs_tile_rows", C.c_int, _vp)
_sig("hs_farm_create", _vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int)
_sig("hs_farm_scene", _vp, _vp, C.c_int)
_sig("hs_farm_size", C.c_int, _vp)
_sig("hs_farm_transport", C.c_char_p, _vp)
_sig("hs_farm_render", C.c_int, _vp, _vp, C.c_int)
_sig("hs_farm_sync", C.c_int, _vp)
_sig("hs_farm_host_us", C.c_double, _vp)
_sig("hs_farm_set_parallel", C.c_int, _vp, C.c_int)
_sig("hs_farm_destroy", None, _vp)
_sig("ptrt_create_interleaved", C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp))
_sig("ptrt_farm_create", C.c_int, C.POINTER(_vp), C.c_int, C.POINTER(_vp))
_sig("ptrt_farm_bands", C.c_int, _vp)
_sig("ptrt_farm_transport", C.c_char_p, _vp)
_sig("ptrt_farm_part_is_local", C.c_int, _vp, C.c_int)
_sig("ptrt_farm_render", C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int)
_sig("ptrt_farm_gather", C.c_int, _vp, _vp, C.c_int)
_sig("ptrt_farm_sync", C.c_int, _vp)
_sig("ptrt_farm_device_frame", _vp, _vp, _vp, C.c_int)
_sig("ptrt_farm_host_us", C.c_double, _vp)
_sig("ptrt_farm_set_option", C.c_int, _vp, C.c_char_p, C.c_longlong)
_sig("ptrt_farm_destroy", None, _vp)
_sig("hs_scene_destroy", None, _vp)
_sig("hs_backend", _vp, _vp)
_sig("hs_init_blue_noise", C.c_int, _vp)
_sig("hs_blue_noise_table", None, _fp)
_sig("hs_blue_noise_generate", None, C.c_int, C.c_int, _fp)
_sig("hs_add_cube", C.c_int, _vp, _fp)
_sig("hs_add_sphere", C.c_int, _vp, C.c_int, _fp)
_sig("hs_add_plane_xz", C.c_int, _vp, C.c_float, C.c_float, _fp)
_sig("hs_add_triangles", C.c_int, _vp, _fp, C.c_int, _fp)
_sig("hs_add_mesh_obj", C.c_int, _vp, C.c_char_p, _fp)
_sig("hs_add_checkerboard", C.c_int, _vp, C.c_float, C.c_int, C.c_float, _fp, _fp)
_sig("hs_mesh_op", C.c_int, _vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float)
_sig("hs_mesh_set_vertices", C.c_int, _vp, C.c_int, _fp, C.c_int)
_sig("hs_mesh_set_triangle_soup", C.c_int, _vp, C.c_int, _fp, C.c_int)
_sig("hs_set_dynamic_geometry_policy", C.c_int, _vp, C.c_int)
_sig("hs_commit_counts", C.c_int, _vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong))
_sig("hs_commit_host_us", C.c_int, _vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
_sig("hs_mesh_counts", C.c_int, _vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig("hs_add_point_light", None, _vp, _fp, _fp, C.c_float, C.c_float, C.c_float)
_sig("hs_add_directional_light", None, _vp, _fp, _fp, C.c_float)
_sig("hs_add_spot_light", None, _vp, _fp, _fp, _fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float)
_sig("hs_move_light_to", None, _vp, C.c_int, _fp)
_sig("hs_set_camera", None, _vp, _fp, _fp, _fp, C.c_float, C.c_float, C.c_float)
_sig("hs_move_camera", None, _vp, _fp)
_sig("hs_look_camera_at", None, _vp, _fp)
_sig("hs_set_sky_gradient", None, _vp, _fp, _fp)
_sig("hs_disable_sky", None, _vp)
_sig("hs_load_hdri", C.c_int, _vp, C.c_char_p)
_sig("hs_set_environment_map", C.c_int, _vp, _fp, C.c_int, C.c_int)
_sig("hs_free_hdri", None, _vp)
_sig("ptrt_set_env_map", C.c_int, _vp, _fp, C.c_int, C.c_int)
_sig("hs_set_bvh_leaf_target", None, _vp, C.c_int, C.c_int)
_sig("hs_set_max_bounce_depth", None, _vp, C.c_int)
_sig("hs_set_samples_per_pixel", None, _vp, C.c_int)
_sig("hs_set_perf_samples_per_pixel", None, _vp, C.c_int)
_sig("hs_set_max_depth", None, _vp, C.c_int)
_sig("hs_get_samples_per_pixel", C.c_int, _vp)
_sig("hs_serialize_scene", C.c_size_t, _vp, _vp, C.c_size_t)
_sig("hs_set_denoiser_enabled", C.c_int, _vp, C.c_int)
_sig("hs_set_bloom_enabled", None, _vp, C.c_int)
_sig("hs_set_performance_preset", C.c_int, _vp, C.c_char_p)
_sig("hs_set_resolution_scale", C.c_int, _vp, C.c_float)
_sig("hs_get_render_size", None, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig("ptrt_set_bloom", C.c_int, _vp, C.c_int)
_sig("ptrt_set_render_size", C.c_int, _vp, C.c_int, C.c_int)
_sig("hs_get_settings", None, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
     C.POINTER(C.c_int), _fp)
_sig("hs_set_mesh_material", C.c_int, _vp, C.c_int, _fp)
_sig("hs_upload", C.c_int, _vp)
_sig("hs_commit_object_changes", C.c_int, _vp)
_sig("hs_get_view_proj", None, _vp, C.c_int, _fp)
_sig("ptrt_denoiser_default_settings", None, _vp)
_sig("ptrt_denoiser_enable", C.c_int, _vp, _vp)
_sig("ptrt_denoiser_disable", C.c_int, _vp)
_sig("ptrt_set_prev_view_proj", C.c_int, _vp, _fp)
_sig("hs_refit_object_changes", C.c_int, _vp)
_sig("hs_refit_from_device", C.c_int, _vp, C.c_int, _vp)
_sig("hs_refit_from_host", C.c_int, _vp, C.c_int, _vp)
_sig("ptrt_update_vertices", C.c_int, _vp, C.c_int, _fp, C.c_int, C.c_int)
_sig("ptrt_refit", C.c_int, _vp)
_sig("ptrt_build_bvh", C.c_int, _vp, C.c_int)
_sig("ptrt_read_prim_order", C.c_int, _vp, C.c_int, C.POINTER(C.c_int), C.c_int)
_sig("ptrt_update_triangles", C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int)
_sig("hs_rebuild_object_changes", C.c_int, _vp, C.c_int)
_sig("hs_rebuild_from_device", C.c_int, _vp, C.c_int, _vp)
_sig("hs_update_triangles", C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int)
_sig("hs_mesh_prim_indices", C.c_int, _vp, C.c_int, C.POINTER(C.c_int), C.c_int)
_sig("hs_render_to_device", C.c_int, _vp, _vp)
_sig("hs_render_to_host", C.c_int, _vp, _vp)
_sig("hs_post_frame", C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int)
_sig("hs_get_frame_count", C.c_int, _vp)
_sig("hs_set_frame_count", None, _vp, C.c_int)
_sig("hs_trace_single_ray", C.c_int, _vp, _fp, _fp, C.POINTER(Hit))
_sig("hs_save_ppm", C.c_int, _vp, C.c_char_p, _vp)
_sig("hs_view_run", C.c_int, _vp, C.c_int, C.c_int, _vp, C.POINTER(C.c_double), C.c_char_p, C.c_int)
_sig("ptrt_present_create", C.c_int, _vp, C.c_int)
_sig("ptrt_present_map", C.c_int, _vp, C.c_int, C.POINTER(_vp))
_sig("ptrt_present_unmap", C.c_int, _vp, C.c_int)
_sig("ptrt_present_acquire", C.c_int, _vp, C.c_int, C.POINTER(_vp))
_sig("ptrt_present_destroy", C.c_int, _vp)
_sig("ptrt_ring_create", C.c_int, C.c_int, C.c_size_t, C.c_int, C.POINTER(_vp))
_sig("ptrt_ring_map", C.c_int, _vp, C.c_int, C.POINTER(_vp))
_sig("ptrt_ring_unmap", C.c_int, _vp, C.c_int)
_sig("ptrt_ring_acquire", C.c_int, _vp, C.c_int, C.POINTER(_vp))
_sig("ptrt_ring_destroy", None, _vp)
_sig("hs_flatten", C.POINTER(SceneDesc), _vp)


class PtrtError(RuntimeError):
    pass


def _f3(v):
    return (C.c_float * 3)(*[float(a) for a in v])


def _fptr(a):
    return a.ctypes.data_as(_fp)


class Material:
    """`struct Material` (scene/material_lib.cuh:12-105) as 27 floats."""
    NAMES = {"albedo": (0, 3), "specular": (3, 3), "metallic": (6, 1), "roughness": (7, 1), "emission": (8, 3),
             "ior": (11, 1), "transmission": (12, 1), "transmissionRoughness": (13, 1), "clearcoat": (14, 1),
             "clearcoatRoughness": (15, 1), "subsurfaceColor": (16, 3), "subsurfaceRadius": (19, 1),
             "anisotropy": (20, 1), "sheen": (21, 1), "sheenTint": (22, 3), "iridescence": (25, 1),
             "iridescenceThickness": (26, 1)}

    def __init__(self, albedo=None, roughness=0.5, metallic=0.0, **fields):
        self.f = np.zeros(27, dtype=np.float32)
        if albedo is None:
            lib.hs_material_default(_fptr(self.f))
        else:
            lib.hs_material_make(_f3(albedo), float(roughness), float(metallic), _fptr(self.f))
        for k, v in fields.items():
            self.set(k, v)

    def set(self, name, value):
        off, n = self.NAMES[name]
        self.f[off:off + n] = np.asarray(value, dtype=np.float32).reshape(-1) if n == 3 else np.float32(value)
        return self

    def get(self, name):
        off, n = self.NAMES[name]
        return self.f[off:off + n].copy() if n == 3 else float(self.f[off])

    def ptr(self):
        return _fptr(self.f)


def blue_noise_table():
    """The 64x64x2 table `initBlueNoise()` uploads (common/bluenoise.cuh:79-198), libstdc++ build."""
    t = np.zeros(64 * 64 * 2, dtype=np.float32)
    lib.hs_blue_noise_table(_fptr(t))
    return t


class Scene:
    """The reference's `Scene` host API; see host/ptrt/scene.hpp for per-method citations."""

    def __init__(self, width, height, tile_y0=0, tile_rows=0, device=0, interleave=None):
        """`interleave=(phase, period)`: the scene renders every period-th 8-row strip of the frame from strip `phase`
        (ptrt_create_interleaved) instead of the contiguous rows [tile_y0, tile_y0 + tile_rows)."""
        self.width, self.height = int(width), int(height)
        self.device = device
        self.interleave = interleave
        if interleave is not None:
            phase, period = interleave
            self._h = lib.hs_scene_create_interleaved(self.width, self.height, int(phase), int(period), device)
            if not self._h:
                raise PtrtError(lib.hs_last_error().decode())
            self.tile_y0, self.tile_rows = int(phase) * 8, lib.hs_tile_rows(self._h)
            return
        self.tile_y0 = int(tile_y0)
        self.tile_rows = int(tile_rows) if tile_rows > 0 else self.height
        self._h = lib.hs_scene_create(self.width, self.height, tile_y0, tile_rows, device)
        if not self._h:
            raise PtrtError(lib.hs_last_error().decode())

    @classmethod
    def _borrowed(cls, handle, width, height, device):
        """A Scene owned by someone else (a C++ TileFarm's part): same methods, never destroyed from here."""
        s = cls.__new__(cls)
        s._h, s._owned = handle, False
        s.width, s.height, s.device, s.interleave = int(width), int(height), device, None
        s.tile_y0, s.tile_rows = 0, lib.hs_tile_rows(handle)
        return s

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                lib.hs_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise PtrtError(lib.hs_last_error().decode())
        return rc

    @property
    def ctx(self):
        return lib.hs_backend(self._h)

    def _cchk(self, rc):
        if rc != PTRT_OK:
            raise PtrtError(lib.ptrt_last_error(self.ctx).decode())

    # geometry
    def addCube(self, mat): return self._chk(lib.hs_add_cube(self._h, mat.ptr()))
    def addSphere(self, segments, mat): return self._chk(lib.hs_add_sphere(self._h, segments, mat.ptr()))
    def addPlaneXZ(self, y, half, mat): return self._chk(lib.hs_add_plane_xz(self._h, y, half, mat.ptr()))

    def addTriangles(self, tris, mat):
        a = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
        return self._chk(lib.hs_add_triangles(self._h, _fptr(a), a.shape[0], mat.ptr()))

    def addMesh(self, path, mat): return self._chk(lib.hs_add_mesh_obj(self._h, path.encode(), mat.ptr()))

    def addCheckerboardPlaneXZ(self, y, tiles, tile_size, white, black):
        self._chk(lib.hs_add_checkerboard(self._h, y, tiles, tile_size, white.ptr(), black.ptr()))

    def _op(self, mesh, op, v): self._chk(lib.hs_mesh_op(self._h, mesh, op, float(v[0]), float(v[1]), float(v[2])))
    def scale(self, mesh, s): self._op(mesh, 0, (s, s, s) if np.isscalar(s) else s)
    def translate(self, mesh, d): self._op(mesh, 1, d)
    def moveTo(self, mesh, p): self._op(mesh, 2, p)
    def rotateSelfEulerXYZ(self, mesh, r): self._op(mesh, 3, r)
    def setPosition(self, mesh, p): self._op(mesh, 4, p)
    def setRotation(self, mesh, r): self._op(mesh, 5, r)
    def setInstanceScale(self, mesh, s): self._op(mesh, 6, s)

    def setVertices(self, mesh, xyz):
        a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self._chk(lib.hs_mesh_set_vertices(self._h, mesh, _fptr(a), a.shape[0]))

    def setTriangleSoup(self, mesh, tris):
        """What the reference's updatePTScene does to a `Triangles` mesh (PTRTtransfer.cuh:2249-2270): vertices and faces
        rewritten, bvhDirty / vertsDirty set, local box recomputed; the caller then commits (commitObjectChanges)."""
        a = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
        self._chk(lib.hs_mesh_set_triangle_soup(self._h, mesh, _fptr(a), a.shape[0]))

    POLICIES = {"HostRebuild": 0, "GpuRefit": 1, "GpuRebuild": 2}

    def setDynamicGeometryPolicy(self, policy):
        self._chk(lib.hs_set_dynamic_geometry_policy(self._h, self.POLICIES.get(policy, policy)))

    def commitCounts(self):
        """(commits that took the GPU refit / rebuild path, geometry uploads) so far."""
        a, b = C.c_longlong(), C.c_longlong()
        lib.hs_commit_counts(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def commitHostMicros(self):
        """(caller, mirror, compare): accumulated host microseconds of setTriangleSoup (what updatePTScene does to a `Triangles`
        mesh), of the commits themselves (updateAccelerationStructures) and, within those, of the face-list comparison."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        lib.hs_commit_host_us(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def meshCounts(self, mesh):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._chk(lib.hs_mesh_counts(self._h, mesh, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # lights / camera / sky
    def addPointLight(self, pos, col, intensity=1.0, range=100.0, radius=0.0):
        lib.hs_add_point_light(self._h, _f3(pos), _f3(col), intensity, range, radius)

    def addDirectionalLight(self, direction, col, intensity=1.0):
        lib.hs_add_directional_light(self._h, _f3(direction), _f3(col), intensity)

    def addSpotLight(self, pos, direction, col, intensity=1.0, inner=0.5, outer=0.7, range=100.0, radius=0.0):
        lib.hs_add_spot_light(self._h, _f3(pos), _f3(direction), _f3(col), intensity, inner, outer, range, radius)

    def moveLightTo(self, i, pos): lib.hs_move_light_to(self._h, i, _f3(pos))

    def setCamera(self, lookfrom, lookat, vup, vfov, aperture=0.0, focus_dist=1.0):
        lib.hs_set_camera(self._h, _f3(lookfrom), _f3(lookat), _f3(vup), vfov, aperture, focus_dist)

    def moveCamera(self, pos): lib.hs_move_camera(self._h, _f3(pos))
    def lookCameraAt(self, at): lib.hs_look_camera_at(self._h, _f3(at))
    def setSkyGradient(self, top, bottom): lib.hs_set_sky_gradient(self._h, _f3(top), _f3(bottom))
    def disableSky(self): lib.hs_disable_sky(self._h)
    def loadHDRI(self, path): self._chk(lib.hs_load_hdri(self._h, str(path).encode()))
    def freeHDRI(self): lib.hs_free_hdri(self._h)

    def setEnvironmentMap(self, rgba):
        """(h, w, 4) float32 equirectangular map, row 0 = v 0 (what loadHDRI would hand over)."""
        a = np.ascontiguousarray(rgba, dtype=np.float32)
        assert a.ndim == 3 and a.shape[2] == 4
        self._chk(lib.hs_set_environment_map(self._h, _fptr(a), a.shape[1], a.shape[0]))

    # settings
    def setBVHLeafTarget(self, target, tol=5): lib.hs_set_bvh_leaf_target(self._h, target, tol)
    def setMaxBounceDepth(self, d): lib.hs_set_max_bounce_depth(self._h, d)
    # Scene::setSamplesPerPixel / setMaxDepth (scene.cuh:1248-1255): stored and IGNORED by render_to_device, as in
    # the reference; the sample count a frame uses is perfSettings.samplesPerPixel -> setPerfSamplesPerPixel
    def setSamplesPerPixel(self, n): lib.hs_set_samples_per_pixel(self._h, n)
    def setMaxDepth(self, d): lib.hs_set_max_depth(self._h, d)
    def getSamplesPerPixel(self): return lib.hs_get_samples_per_pixel(self._h)
    def setPerfSamplesPerPixel(self, n): lib.hs_set_perf_samples_per_pixel(self._h, n)
    def setDenoiserEnabled(self, e): self._chk(lib.hs_set_denoiser_enabled(self._h, int(e)))
    def setBloomEnabled(self, e): lib.hs_set_bloom_enabled(self._h, int(e))
    def setPerformancePreset(self, name): self._chk(lib.hs_set_performance_preset(self._h, name.encode()))
    def setResolutionScale(self, s): self._chk(lib.hs_set_resolution_scale(self._h, float(s)))

    def renderSize(self):
        """(render_width, render_height): the size the path tracer runs at (perfSettings.resolutionScale)."""
        w, h = C.c_int(), C.c_int()
        lib.hs_get_render_size(self._h, C.byref(w), C.byref(h))
        return w.value, h.value
    def setMeshMaterial(self, mesh, mat): self._chk(lib.hs_set_mesh_material(self._h, mesh, mat.ptr()))

    def settings(self):
        spp, depth, dn, bl, sc = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_float()
        lib.hs_get_settings(self._h, C.byref(spp), C.byref(depth), C.byref(dn), C.byref(bl), C.byref(sc))
        return dict(spp=spp.value, depth=depth.value, denoiser=bool(dn.value), bloom=bool(bl.value), scale=sc.value)

    # upload / render
    def initBlueNoise(self): self._chk(lib.hs_init_blue_noise(self._h))
    def uploadToGPU(self): self._chk(lib.hs_upload(self._h))
    def commitObjectChanges(self): self._chk(lib.hs_commit_object_changes(self._h))

    def view_proj(self, current=False):
        """proj*view (16 floats, column-major as mat4 stores them): previous frame's by default."""
        m = np.zeros(16, dtype=np.float32)
        lib.hs_get_view_proj(self._h, 1 if current else 0, _fptr(m))
        return m

    def refitObjectChanges(self):
        """Dynamic vertices, same topology: host + GPU BVH refit instead of the reference's rebuild."""
        self._chk(lib.hs_refit_object_changes(self._h))

    def refitFromDevice(self, mesh, device_ptr):
        """New vertex positions (n x 3 float32) already in device memory -> update + GPU refit, no host sync."""
        self._chk(lib.hs_refit_from_device(self._h, mesh, C.c_void_p(device_ptr)))

    def refitFromHost(self, mesh, host_ptr):
        """New positions from HOST memory (a raw address: pinned memory makes the copy asynchronous) + GPU refit."""
        self._chk(lib.hs_refit_from_host(self._h, mesh, C.c_void_p(host_ptr)))

    def rebuildObjectChanges(self, sync_host_copy=True):
        """commitObjectChanges for unchanged face counts with the BVH rebuilt on the GPU (ptrt_build_bvh)."""
        self._chk(lib.hs_rebuild_object_changes(self._h, int(sync_host_copy)))

    def rebuildFromDevice(self, mesh, device_ptr):
        """New vertex positions already in device memory -> update + GPU BVH rebuild, no host sync."""
        self._chk(lib.hs_rebuild_from_device(self._h, mesh, C.c_void_p(device_ptr)))

    def updateTriangles(self, mesh, verts9, tri_count=None, device_ptr=None):
        """A new triangle list (<= the uploaded count) for a soup mesh + GPU rebuild: numpy (n,9) or a device pointer."""
        if device_ptr is not None:
            self._chk(lib.hs_update_triangles(self._h, mesh, C.c_void_p(device_ptr), int(tri_count), 1))
            return
        a = np.ascontiguousarray(verts9, dtype=np.float32).reshape(-1, 9)
        self._chk(lib.hs_update_triangles(self._h, mesh, a.ctypes.data_as(_vp), a.shape[0], 0))

    def primIndices(self, mesh):
        """Host copy of the mesh's `primIndices` (mesh.cuh:57)."""
        n = self.meshCounts(mesh)[1]
        out = np.zeros(n, dtype=np.int32)
        self._chk(lib.hs_mesh_prim_indices(self._h, mesh, out.ctypes.data_as(C.POINTER(C.c_int)), n))
        return out

    def getFrameCount(self): return lib.hs_get_frame_count(self._h)
    def setFrameCount(self, f): lib.hs_set_frame_count(self._h, f)

    def render_to_device(self, device_ptr):
        """`Scene::render_to_device(unsigned char*)`: asynchronous, RGB8 bottom-up into device memory."""
        self._chk(lib.hs_render_to_device(self._h, C.c_void_p(device_ptr)))

    def render_to_host(self):
        out = np.empty((self.tile_rows, self.width, 3), dtype=np.uint8)
        self._chk(lib.hs_render_to_host(self._h, out.ctypes.data_as(_vp)))
        return out

    def post_frame(self, accum_ptr, normal_ptr, depth_ptr, object_id_ptr, out_device_ptr=None):
        """Presenting rank of the tile farm: motion vectors + denoiser + bloom + tonemap of this full-frame scene
        over a gathered frame (device pointers, top-down; `ptrt_post_frame`).  RGB8 goes to `out_device_ptr`, or is
        returned as a host array."""
        if out_device_ptr is not None:
            self._chk(lib.hs_post_frame(self._h, accum_ptr, normal_ptr, depth_ptr, object_id_ptr, C.c_void_p(out_device_ptr), 1))
            return None
        out = np.empty((self.tile_rows, self.width, 3), dtype=np.uint8)
        self._chk(lib.hs_post_frame(self._h, accum_ptr, normal_ptr, depth_ptr, object_id_ptr, out.ctypes.data_as(_vp), 0))
        return out

    def view_run(self, frames, slots=2, keep=True, dump_prefix="", dump_every=0):
        """The reference's viewer loop (map_pbo -> render_to_device -> unmap -> blit -> draw) over the HIP
        presentation ring, headless.  Returns (frames as uint8 (n, H, W, 3) or None, wall ms per frame)."""
        out = np.empty((frames, self.tile_rows, self.width, 3), dtype=np.uint8) if keep else None
        ms = C.c_double()
        self._chk(lib.hs_view_run(self._h, frames, slots, out.ctypes.data_as(_vp) if keep else None, C.byref(ms),
                                  dump_prefix.encode(), dump_every))
        return out, ms.value

    def sync(self): self._cchk(lib.ptrt_sync(self.ctx))

    def device_array(self, kind):
        """Zero-copy view of a frame buffer in device memory for array libraries that accept
        `__cuda_array_interface__` (e.g. `torch.as_tensor(scene.device_array(P.BUF_ACCUM), device="cuda")`)."""
        n = self.tile_rows * self.width
        shape, typestr = {BUF_ACCUM: ((n, 3), "<f4"), BUF_NORMAL: ((n, 3), "<f4"), BUF_DEPTH: ((n, 1), "<f4"),
                          BUF_OBJECT_ID: ((n, 1), "<i4")}[kind]
        ptr = lib.ptrt_device_buffer(self.ctx, kind)
        if not ptr:
            raise PtrtError(lib.ptrt_last_error(self.ctx).decode())

        class _View:
            __cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}
        return _View()

    def read(self, kind):
        n = self.tile_rows * self.width
        rw, rh = self.renderSize()
        r = n if (rw, rh) == (self.width, self.height) else rw * rh   # render-size images (see ptrt.h)
        shape, dt = {BUF_ACCUM: ((n, 3), np.float32), BUF_NORMAL: ((r, 3), np.float32), BUF_DEPTH: ((r,), np.float32),
                     BUF_OBJECT_ID: ((r,), np.int32), BUF_RGB8: ((self.tile_rows, self.width, 3), np.uint8),
                     BUF_RNG: ((n, 6), np.uint32), BUF_DENOISED: ((r, 3), np.float32),
                     BUF_MOTION: ((r, 2), np.float32), BUF_RENDER_ACCUM: ((r, 3), np.float32)}[kind]
        out = np.empty(shape, dtype=dt)
        self._cchk(lib.ptrt_read_buffer(self.ctx, kind, out.ctypes.data_as(_vp), out.nbytes))
        return out

    def write_rng(self, states):
        a = np.ascontiguousarray(states, dtype=np.uint32)
        self._cchk(lib.ptrt_write_rng(self.ctx, a.ctypes.data_as(C.POINTER(C.c_uint32)), a.nbytes))

    def reset_rng(self, seed=DEFAULT_SEED): self._cchk(lib.ptrt_reset_rng(self.ctx, seed))
    def set_option(self, name, value): self._cchk(lib.ptrt_set_option(self.ctx, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_longlong(0)
        self._cchk(lib.ptrt_get_option(self.ctx, name.encode(), C.byref(v)))
        return v.value

    def stats(self):
        s = Stats()
        self._cchk(lib.ptrt_get_stats(self.ctx, C.byref(s)))
        return dict(extension_rays=s.extension_rays, shadow_rays=s.shadow_rays, paths=s.paths,
                    shadow_rays_walked=s.shadow_rays_walked)

    def last_kernel_ms(self):
        a = C.c_float()
        self._cchk(lib.ptrt_last_kernel_ms(self.ctx, C.byref(a), None))
        return a.value

    def set_stream(self, hip_stream):
        self._cchk(lib.ptrt_set_stream(self.ctx, C.c_void_p(hip_stream)))

    def kernel_ms_history(self, max_n=256):
        out = np.zeros(max_n, dtype=np.float32)
        n = lib.ptrt_kernel_ms_history(self.ctx, _fptr(out), max_n)
        if n < 0:
            raise PtrtError(lib.ptrt_last_error(self.ctx).decode())
        return out[:n]

    def launch_ms_history(self, max_n=1024):
        """(trace_ms, tail_ms) of the launches of the last overlapping frames (option time_launches), oldest first."""
        a, b = np.zeros(max_n, dtype=np.float32), np.zeros(max_n, dtype=np.float32)
        n = lib.ptrt_launch_ms_history(self.ctx, _fptr(a), _fptr(b), max_n)
        if n < 0:
            raise PtrtError(lib.ptrt_last_error(self.ctx).decode())
        return a[:n], b[:n]

    def trace_rays(self, origins, directions):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(o.shape[0], dtype=HIT_DTYPE)
        self._cchk(lib.ptrt_trace_rays(self.ctx, _fptr(o), _fptr(d), o.shape[0], out.ctypes.data_as(_vp)))
        return out

    def traceSingleRay(self, origin, direction):
        h = Hit()
        self._chk(lib.hs_trace_single_ray(self._h, _f3(origin), _f3(direction), C.byref(h)))
        return h

    def saveAsPPM(self, path, pixels):
        a = np.ascontiguousarray(pixels, dtype=np.uint8)
        self._chk(lib.hs_save_ppm(self._h, path.encode(), a.ctypes.data_as(_vp)))

    def serialize(self):
        """Canonical byte stream of the flattened scene (host/ptrt/serialize.hpp)."""
        n = lib.hs_serialize_scene(self._h, None, 0)
        if not n:
            raise PtrtError(lib.hs_last_error().decode())
        buf = (C.c_ubyte * n)()
        lib.hs_serialize_scene(self._h, buf, n)
        return bytes(buf)

    def flatten(self):
        """Pointer to the flattened `ptrt_scene_desc` (host arrays owned by the C++ Scene)."""
        p = lib.hs_flatten(self._h)
        if not p:
            raise PtrtError(lib.hs_last_error().decode())
        return p


class TileFarm:
    """host/ptrt/farm.hpp from Python: one C++ Scene per entry of `devices` (bands, or interleaved 8-row strips),
    their images gathered below the C ABI (ptrt_farm_*: device copies on the presenting GPU, RCCL from the others)."""

    def __init__(self, width, height, devices, strips=False):
        self.width, self.height = int(width), int(height)
        arr = (C.c_int * len(devices))(*devices)
        self._f = lib.hs_farm_create(self.width, self.height, arr, len(devices), int(bool(strips)))
        if not self._f:
            raise PtrtError(lib.hs_last_error().decode())
        self.scenes = [Scene._borrowed(lib.hs_farm_scene(self._f, i), width, height, devices[i]) for i in range(len(devices))]

    @property
    def transport(self):
        return lib.hs_farm_transport(self._f).decode()

    def render_to_device(self, ptr):
        if lib.hs_farm_render(self._f, ptr, 1) < 0:
            raise PtrtError(lib.hs_last_error().decode())

    def render_to_host(self):
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        if lib.hs_farm_render(self._f, out.ctypes.data_as(_vp), 0) < 0:
            raise PtrtError(lib.hs_last_error().decode())
        return out

    @property
    def host_us(self):
        """Host time (us) the calling thread spent inside the last render (every part's enqueue + the gather's calls)."""
        return float(lib.hs_farm_host_us(self._f))

    def set_parallel(self, on):
        if lib.hs_farm_set_parallel(self._f, int(bool(on))) < 0:
            raise PtrtError(lib.hs_last_error().decode())

    def sync(self):
        if lib.hs_farm_sync(self._f) < 0:
            raise PtrtError(lib.hs_last_error().decode())

    def close(self):
        if getattr(self, "_f", None):
            for s in self.scenes:
                s.close()
            lib.hs_farm_destroy(self._f)
            self._f = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


from . import scenes  # noqa: E402,F401

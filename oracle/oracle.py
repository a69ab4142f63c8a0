"""ctypes binding of the CPU oracle (oracle/libptrt_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  PARITY UNPINNED (see ptrt_oracle.cpp header).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PTRT_ORACLE_LIB: another build of the same sources -- tools/fmad_sensitivity.py loads the one compiled with every a*b+c
# contracted, to estimate how far a CUDA build with nvcc's default -fmad=true may sit from this one)
LIB_PATH = os.environ.get("PTRT_ORACLE_LIB") or os.path.join(_HERE, "libptrt_oracle.so")


def build():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", _HERE])


if not os.path.exists(LIB_PATH):
    build()
lib = C.CDLL(LIB_PATH)


class RenderArgs(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("tile_y0", C.c_int32), ("tile_rows", C.c_int32),
                ("spp", C.c_int32), ("max_depth", C.c_int32), ("frame_count", C.c_int32), ("threads", C.c_int32),
                ("blue_noise", C.POINTER(C.c_float)), ("rng", C.POINTER(C.c_uint32)),
                ("accum", C.POINTER(C.c_float)), ("normal", C.POINTER(C.c_float)), ("depth", C.POINTER(C.c_float)),
                ("object_id", C.POINTER(C.c_int32)),
                ("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("paths", C.c_uint64),
                ("shadow_rays_walked", C.c_uint64)]


_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint32)
lib.oracle_has_fma.restype = C.c_int
lib.oracle_xorwow_init.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_ulonglong, _up]
lib.oracle_xorwow_custom.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_int, _up, _up]
lib.oracle_xorwow_draw.argtypes = [_up, C.c_int, _up, _fp]
lib.oracle_render.argtypes = [C.c_void_p, C.POINTER(RenderArgs)]
lib.oracle_render.restype = C.c_int
lib.oracle_tonemap.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
lib.oracle_trace_rays.argtypes = [C.c_void_p, _fp, _fp, C.c_int, C.c_void_p]
lib.oracle_any_hit.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_int, C.POINTER(C.c_int32)]
lib.oracle_taa_jitter.argtypes = [C.c_int, _fp]
lib.oracle_detmath.argtypes = [C.c_int, _fp, _fp, C.c_int, _fp]
lib.oracle_eval_bsdf.argtypes = [C.c_void_p, C.c_int, _fp, _fp, _fp, C.c_int, _fp, _fp]
lib.oracle_scatter.argtypes = [C.c_void_p, C.c_int, _fp, _fp, C.c_int, _up, _fp]

HIT_DTYPE = np.dtype([("hit", "<i4"), ("t", "<f4"), ("point", "<f4", 3), ("normal", "<f4", 3),
                      ("mesh_index", "<i4"), ("front_face", "<i4"), ("u", "<f4"), ("v", "<f4"),
                      ("face_index", "<i4"), ("local_point", "<f4", 3)])


def _f(a):
    return a.ctypes.data_as(_fp)


def _u(a):
    return a.ctypes.data_as(_up)


def xorwow_init(seed, first, count):
    out = np.zeros((count, 6), dtype=np.uint32)
    lib.oracle_xorwow_init(seed, first, count, _u(out))
    return out


def xorwow_custom(seed, subsequence, consts, n_draws):
    draws = np.zeros(max(n_draws, 1), dtype=np.uint32)
    state = np.zeros(6, dtype=np.uint32)
    lib.oracle_xorwow_custom(seed, subsequence, consts[0], consts[1], consts[2], consts[3], n_draws, _u(draws),
                             _u(state))
    return draws[:n_draws], state


def xorwow_draw(state6, n, uniform=False):
    st = np.array(state6, dtype=np.uint32).copy()
    if uniform:
        out = np.zeros(n, dtype=np.float32)
        lib.oracle_xorwow_draw(_u(st), n, None, _f(out))
    else:
        out = np.zeros(n, dtype=np.uint32)
        lib.oracle_xorwow_draw(_u(st), n, _u(out), None)
    return out, st


def render(scene_desc_ptr, width, height, spp, max_depth, frame_count, blue_noise, rng, tile_y0=0, tile_rows=0,
           threads=1):
    """path_trace_kernel on the CPU.  `rng` (rows*W,6) uint32 is advanced in place.
    Returns dict(accum, normal, depth, object_id, stats)."""
    rows = tile_rows if tile_rows > 0 else height
    n = rows * width
    assert rng.shape == (n, 6) and rng.dtype == np.uint32 and rng.flags.c_contiguous
    bn = np.ascontiguousarray(blue_noise, dtype=np.float32)
    out = dict(accum=np.zeros((n, 3), np.float32), normal=np.zeros((n, 3), np.float32),
               depth=np.zeros(n, np.float32), object_id=np.zeros(n, np.int32))
    a = RenderArgs(width, height, tile_y0, rows, spp, max_depth, frame_count, threads, _f(bn), _u(rng),
                   _f(out["accum"]), _f(out["normal"]), _f(out["depth"]),
                   out["object_id"].ctypes.data_as(C.POINTER(C.c_int32)), 0, 0, 0, 0)
    rc = lib.oracle_render(C.cast(scene_desc_ptr, C.c_void_p), C.byref(a))
    if rc != 0:
        raise RuntimeError(f"oracle_render failed ({rc}); -2 means the CPU lacks FMA")
    out["stats"] = dict(extension_rays=a.extension_rays, shadow_rays=a.shadow_rays, paths=a.paths,
                        shadow_rays_walked=a.shadow_rays_walked)
    return out


def tonemap(accum, width, rows, threads=1):
    acc = np.ascontiguousarray(accum, dtype=np.float32)
    out = np.zeros((rows, width, 3), dtype=np.uint8)
    lib.oracle_tonemap(_f(acc), width, rows, 1, out.ctypes.data_as(C.c_void_p), threads)
    return out


def trace_rays(scene_desc_ptr, origins, directions):
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(o.shape[0], dtype=HIT_DTYPE)
    lib.oracle_trace_rays(C.cast(scene_desc_ptr, C.c_void_p), _f(o), _f(d), o.shape[0],
                          out.ctypes.data_as(C.c_void_p))
    return out


def any_hit(scene_desc_ptr, origins, directions, tmax):
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tmax, dtype=np.float32)
    out = np.zeros(o.shape[0], dtype=np.int32)
    lib.oracle_any_hit(C.cast(scene_desc_ptr, C.c_void_p), _f(o), _f(d), _f(t), o.shape[0],
                       out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


class DenoiserSettings(C.Structure):
    """DenoiserSettings, non-split subset (denoiser.cuh:36-73); defaults = the diffuse_* channel."""
    _fields_ = [("tau", C.c_float), ("min_alpha", C.c_float), ("max_history", C.c_float),
                ("sigma_luminance", C.c_float), ("sigma_normal", C.c_float), ("sigma_depth", C.c_float),
                ("atrous_iterations", C.c_int32), ("clamp_scale", C.c_float), ("firefly_threshold", C.c_float),
                ("depth_reject_absolute", C.c_float), ("depth_reject_relative", C.c_float),
                ("normal_reject_threshold", C.c_float), ("sky_depth_threshold", C.c_float),
                ("edge_depth_threshold", C.c_float), ("edge_normal_threshold", C.c_float),
                ("use_object_ids", C.c_int32), ("enable_firefly_suppression", C.c_int32)]

    def __init__(self, **kw):
        super().__init__(0.06, 0.05, 32.0, 4.0, 64.0, 0.5, 5, 1.2, 3.0, 0.1, 0.005, 0.95, 1e9, 0.01, 0.95, 1, 1)
        for k, v in kw.items():
            setattr(self, k, v)


class _DenoiserState(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("first_frame", C.c_int32),
                ("history_mean", _fp), ("history_m2", _fp), ("history_length", _fp),
                ("history_normal", _fp), ("history_depth", _fp), ("history_object_id", C.POINTER(C.c_int32))]


lib.oracle_motion_vectors.argtypes = [_fp, C.c_int, C.c_int, _fp, _fp, _fp]
lib.oracle_denoise.argtypes = [C.POINTER(DenoiserSettings), C.POINTER(_DenoiserState), _fp, _fp, _fp, _fp,
                               C.POINTER(C.c_int32), _fp]


class Denoiser:
    """`class Denoiser` (denoiser.cuh:781-1070), non-split path, state held in numpy arrays."""

    def __init__(self, width, height, settings=None):
        self.w, self.h = width, height
        self.settings = settings or DenoiserSettings()
        n = width * height
        self.hmean, self.hm2 = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        self.hlen, self.hdepth = np.zeros(n, np.float32), np.zeros(n, np.float32)
        self.hnormal, self.hobj = np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        self.state = _DenoiserState(width, height, 1, _f(self.hmean), _f(self.hm2), _f(self.hlen), _f(self.hnormal),
                                    _f(self.hdepth), self.hobj.ctypes.data_as(C.POINTER(C.c_int32)))

    def denoise(self, color, normal, depth, motion, object_id):
        color, normal = np.ascontiguousarray(color, np.float32), np.ascontiguousarray(normal, np.float32)
        depth, motion = np.ascontiguousarray(depth, np.float32), np.ascontiguousarray(motion, np.float32)
        oid = np.ascontiguousarray(object_id, np.int32)
        out = np.zeros((self.w * self.h, 3), np.float32)
        lib.oracle_denoise(C.byref(self.settings), C.byref(self.state), _f(color), _f(normal), _f(depth), _f(motion),
                           oid.ctypes.data_as(C.POINTER(C.c_int32)), _f(out))
        return out


def motion_vectors(depth, width, height, camera, prev_view_proj):
    """motion_vector_kernel; `camera` = ptrt_camera-like (origin, lower_left_corner, horizontal, vertical, u, v,
    lens_radius)."""
    cam = np.array([camera.origin.x, camera.origin.y, camera.origin.z,
                    camera.lower_left_corner.x, camera.lower_left_corner.y, camera.lower_left_corner.z,
                    camera.horizontal.x, camera.horizontal.y, camera.horizontal.z,
                    camera.vertical.x, camera.vertical.y, camera.vertical.z,
                    camera.u.x, camera.u.y, camera.u.z, camera.v.x, camera.v.y, camera.v.z, camera.lens_radius],
                   np.float32)
    d = np.ascontiguousarray(depth, np.float32)
    pvp = np.ascontiguousarray(prev_view_proj, np.float32)
    out = np.zeros((width * height, 2), np.float32)
    lib.oracle_motion_vectors(_f(d), width, height, _f(cam), _f(pvp), _f(out))
    return out


lib.oracle_bloom.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int]
lib.oracle_bloom.restype = C.c_int
lib.oracle_upscale.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int]


def bloom(image, width, height, alloc_w=None, alloc_h=None):
    """Step 5 of Scene::render_to_device (bright pass, 6-level blur/downsample chain, upsample-adds) on a
    (height*width, 3) image; returns the image with bloom added."""
    img = np.ascontiguousarray(image, np.float32).copy()
    rc = lib.oracle_bloom(_f(img), width, height, alloc_w or width, alloc_h or height)
    if rc != 0:
        raise RuntimeError("oracle_bloom: a mip level has zero size (the reference dereferences NULL there)")
    return img


def upscale(image, out_w, out_h, in_w, in_h):
    """upscale_bilinear_kernel."""
    src = np.ascontiguousarray(image, np.float32)
    out = np.zeros((out_w * out_h, 3), np.float32)
    lib.oracle_upscale(_f(out), _f(src), out_w, out_h, in_w, in_h)
    return out


def mat3_mul(m9, v3):
    """mat3 * vec3 as the oracle's aces_tonemap forms it (matrix.cuh:35-39)."""
    m = np.ascontiguousarray(m9, dtype=np.float32).reshape(9)
    v = np.ascontiguousarray(v3, dtype=np.float32).reshape(3)
    out = np.zeros(3, np.float32)
    lib.oracle_mat3_mul(_f(m), _f(v), _f(out))
    return out


def taa_jitter(frame_index):
    out = np.zeros(2, np.float32)
    lib.oracle_taa_jitter(int(frame_index), _f(out))
    return out


def detmath(op, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = x if y is None else np.ascontiguousarray(y, dtype=np.float32)
    out = np.zeros_like(x)
    lib.oracle_detmath(op, _f(x), _f(y), x.size, _f(out))
    return out

"""Regenerates tests/golden/function_kats.json: scalar known answers of the shading functions one by one (SURVEY 8(c) golden item 7:
rows a11 evaluateBSDF, a12 material_pdf, a13 material_scatter) for every material of the reference application's library
(app_utils.cuh:60-191), produced by the ORACLE's per-function entry points (oracle_eval_bsdf / oracle_scatter).  Like
oracle_*.npz they pin this repository's restatement against regressions and localise a failure of the whole-frame parity
tests to one function; they are not outputs of the CUDA reference (unbuildable here).  Floats are stored as
their 32-bit patterns.      python tests/golden/make_function_kats.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
NAMES = ["Silver", "BrushedAluminum", "Gold", "Copper", "Titanium", "Glass", "FrostedGlass", "Water", "Diamond", "SoapBubble",
         "OilSlick", "VelvetRed", "SatinBlue", "CarPaintMidnight", "LacqueredWood", "PlasticRed", "RubberBlack", "Wax", "Jade",
         "MarbleCarrara"]
N_EVAL, N_SCATTER = 6, 6


def library_scene(P):
    """One cube per library material (+ an emitter): the material SoA the functions read is the scene's."""
    s = P.Scene(64, 64, device=P.HOST_ONLY)
    for n in NAMES:
        s.addCube(getattr(P.scenes.Materials, n)())
    s.addCube(P.scenes.Materials.GlowingNeon((0.2, 1.0, 0.2)))
    return s


def inputs():
    """Seeded directions: unit N; V and L in N's upper hemisphere mostly, one grazing and one below it; both faces."""
    rs = np.random.RandomState(20261005)
    def unit(v):
        return (v / np.linalg.norm(v)).astype(np.float32)
    ev, sc = [], []
    for m in range(len(NAMES) + 1):
        for k in range(N_EVAL):
            n = unit(rs.normal(size=3))
            t = unit(np.cross(n, rs.normal(size=3)))
            b = unit(np.cross(n, t))
            def hemi(z):
                a = rs.uniform(0, 2 * np.pi)
                r = np.sqrt(max(0.0, 1 - z * z))
                return unit(n * z + t * r * np.cos(a) + b * r * np.sin(a))
            zv = [0.9, 0.6, 0.3, 0.05, 0.7, 0.5][k]
            zl = [0.8, 0.4, 0.7, 0.6, -0.3, 0.02][k]
            v, l = hemi(zv), hemi(zl)
            if k < 2:  # at / next to the mirror direction: the specular lobes of the sharp materials are not zero there
                l = unit(2.0 * float(np.dot(n, v)) * n - v + (0.0 if k == 0 else 0.03) * t)
            ev.append((m, n, v, l, 1 if k % 3 else 0))
        for k in range(N_SCATTER):
            n = unit(rs.normal(size=3))
            t = unit(np.cross(n, rs.normal(size=3)))
            b = unit(np.cross(n, t))
            z = [0.95, 0.5, 0.2, 0.05, 0.7, 0.35][k]
            a = rs.uniform(0, 2 * np.pi)
            r = np.sqrt(1 - z * z)
            rd = unit(-(n * z + t * r * np.cos(a) + b * r * np.sin(a)))  # the ray comes IN: V = -ray_dir lies above the surface
            sc.append((m, n, rd, 1 if k % 2 == 0 else 0, (12345, 7 + 13 * m + k)))
    return ev, sc


def generate(P, O):
    import ctypes as C
    s = library_scene(P)
    mats = C.byref(s.flatten().contents.materials)
    ev, sc = inputs()
    fp = lambda a: np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.POINTER(C.c_float))
    out = {"materials": NAMES + ["GlowingNeon"], "eval": [], "scatter": []}
    for m, n, v, l, ff in ev:
        f3, pdf = np.zeros(3, np.float32), np.zeros(1, np.float32)
        O.lib.oracle_eval_bsdf(mats, m, fp(n), fp(v), fp(l), ff, fp(f3), fp(pdf))
        out["eval"].append({"mat": m, "in": [int(x) for x in np.concatenate([n, v, l]).view(np.uint32)], "front_face": ff,
                            "f": [int(x) for x in f3.view(np.uint32)], "pdf": int(pdf.view(np.uint32)[0])})
    for m, n, rd, ff, (seed, sub) in sc:
        st = O.xorwow_init(seed, sub, 1).reshape(-1).astype(np.uint32).copy()
        st0 = st.copy()
        o8 = np.zeros(8, np.float32)
        O.lib.oracle_scatter(mats, m, fp(n), fp(rd), ff, st.ctypes.data_as(C.POINTER(C.c_uint32)), fp(o8))
        out["scatter"].append({"mat": m, "in": [int(x) for x in np.concatenate([n, rd]).view(np.uint32)], "front_face": ff,
                               "state": [int(x) for x in st0], "dir": [int(x) for x in o8[:3].view(np.uint32)],
                               "att": [int(x) for x in o8[3:6].view(np.uint32)], "pdf": int(o8[6:7].view(np.uint32)[0]),
                               "flags": int(o8[7]), "state_after": [int(x) for x in st]})
    s.close()
    return out


if __name__ == "__main__":
    sys.path[:0] = [os.path.join(ROOT, "ptrt-game-engine_amd"), os.path.join(ROOT, "oracle")]
    import oracle as O
    import ptrt_amd as P
    d = generate(P, O)
    json.dump(d, open(os.path.join(HERE, "function_kats.json"), "w"), indent=0, separators=(",", ":"))
    print(len(d["eval"]), "evaluateBSDF / material_pdf answers,", len(d["scatter"]), "material_scatter answers")

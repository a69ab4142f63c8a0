#!/usr/bin/env python3
"""How far may a CUDA build of the reference sit from this repository's arithmetic?  (nvcc's default
-fmad=true contracts every a*b+c of the reference into a fused multiply-add; oracle and kernels contract only dot / cross /
length^2 and round everything else twice -- "radiance differs from a real CUDA run by ulps; nobody here can say how many".)
This tool SAYS how many, for the contraction part of that difference: it renders the same frames with the oracle as built
(-ffp-contract=off) and with the same sources built -ffp-contract=fast (`make -C oracle fmad`: gcc then fuses a*b+c within an
expression, which is the rule nvcc follows) and compares them pixel by pixel.  It cannot see CUDA's libm (sinf / expf / powf
differ from the deterministic polynomials by ulps as well) or cuRAND's seed constants.  --variant=libm: the same
comparison with the platform's libm in place of the deterministic polynomials (what ANOTHER libm, such as CUDA's, does to the
frame); --variant=fmad_libm: both.  CPU only.
   python tools/fmad_sensitivity.py [scene] [W H spp depth] [--variant=fmad|libm|fmad_libm]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def render(lib, scene, W, H, spp, depth, out):
    code = f"""
import sys, numpy as np
sys.path[:0] = [{os.path.join(ROOT, 'ptrt-game-engine_amd')!r}, {os.path.join(ROOT, 'oracle')!r}]
import oracle as O, ptrt_amd as P
s = P.Scene({W}, {H}, device=P.HOST_ONLY)
getattr(P.scenes, {scene!r})(s) if {scene!r} != 'showcase' else P.scenes.showcase(s, segments=24)
rng = O.xorwow_init(P.DEFAULT_SEED, 0, {W} * {H})
fr = []
for f in range(2):
    r = O.render(s.flatten(), {W}, {H}, {spp}, {depth}, f, P.blue_noise_table(), rng, threads=8)
    fr.append(r)
np.savez({out!r}, accum0=fr[0]['accum'], accum1=fr[1]['accum'], depth=fr[1]['depth'], oid=fr[1]['object_id'], normal=fr[1]['normal'],
         rgb=O.tonemap(fr[1]['accum'], {W}, {H}), rng=rng)
"""
    env = dict(os.environ)
    if lib:
        env["PTRT_ORACLE_LIB"] = lib
    subprocess.check_call([sys.executable, "-c", code], env=env)
    return np.load(out)


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--variant")]
    variant = ([a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--variant=")] or ["fmad"])[0]  # fmad | libm | fmad_libm
    scene = argv[0] if len(argv) > 0 else "cornell"
    W, H, spp, depth = (int(v) for v in argv[1:5]) if len(argv) > 4 else (256, 256, 4, 4)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all", "fmad", "libm"])
    with tempfile.TemporaryDirectory() as d:
        a = render(None, scene, W, H, spp, depth, os.path.join(d, "a.npz"))
        b = render(os.path.join(ROOT, "oracle", f"libptrt_oracle_{variant}.so"), scene, W, H, spp, depth, os.path.join(d, "b.npz"))
        out = {"variant": variant, "scene": scene, "frame": f"{W}x{H} {spp} spp {depth} bounces, second of two consecutive frames", "pixels": W * H}
        out["object_id_differs_px"] = int((a["oid"] != b["oid"]).sum())
        out["depth_bits_differ_px"] = int((a["depth"].view(np.uint32) != b["depth"].view(np.uint32)).sum())
        dd = np.abs(a["depth"].astype(np.float64) - b["depth"]) / np.maximum(np.abs(a["depth"].astype(np.float64)), 1e-30)
        out["depth_rel_diff_max"] = float(dd[np.isfinite(dd)].max())
        out["generator_state_differs_px"] = int((a["rng"] != b["rng"]).any(axis=1).sum())  # a different number of draws: the PATH differed
        x, y = a["accum1"].astype(np.float64), b["accum1"].astype(np.float64)
        l2 = np.linalg.norm(x - y, axis=1)
        rel = l2 / np.maximum(np.linalg.norm(x, axis=1), 1e-3)
        same_path = ~(a["rng"] != b["rng"]).any(axis=1)
        out["radiance_bits_differ_px"] = int((a["accum1"].view(np.uint32) != b["accum1"].view(np.uint32)).any(axis=1).sum())
        q = lambda v, p: float(np.quantile(v, p))
        out["per_pixel_rel_L2 (all pixels)"] = {"median": q(rel, 0.5), "p99": q(rel, 0.99), "max": float(rel.max())}
        if same_path.any():
            r2 = rel[same_path]
            out["per_pixel_rel_L2 (pixels whose paths drew the same number of random numbers)"] = {
                "pixels": int(same_path.sum()), "median": q(r2, 0.5), "p99": q(r2, 0.99), "max": float(r2.max())}
        out["rgb8_bytes_differ"] = int((a["rgb"] != b["rgb"]).sum())
        out["rgb8_max_abs_diff"] = int(np.abs(a["rgb"].astype(np.int16) - b["rgb"]).max())
        out["image_mean_rel_diff"] = float(abs(x.mean() - y.mean()) / x.mean())
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

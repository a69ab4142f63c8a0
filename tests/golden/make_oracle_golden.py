"""Regenerates tests/golden/oracle_*.npz: small frames rendered by the CPU oracle.  They pin the
ORACLE against regressions (they are produced by this repository's own restatement, not by the
CUDA reference, which cannot be built here -- see DESIGN.md "Oracle").
    python tests/golden/make_oracle_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CASES = {
    "cornell_64x64_1spp_d4_f0": ("cornell", 64, 64, 1, 4, 0),
    "cornell_64x64_4spp_d2_f3": ("cornell", 64, 64, 4, 2, 3),
    "showcase12_64x48_2spp_d5_f0": ("showcase12", 64, 48, 2, 5, 0),
}


def render_case(P, O, blue_noise, name):
    scene, w, h, spp, depth, frame = CASES[name]
    s = P.Scene(w, h, device=P.HOST_ONLY)
    if scene == "cornell":
        P.scenes.cornell(s)
    else:
        P.scenes.showcase(s, segments=12)
    rng = O.xorwow_init(12345, 0, w * h)
    r = O.render(s.flatten(), w, h, spp, depth, frame, blue_noise, rng, threads=4)
    r["rgb8"] = O.tonemap(r["accum"], w, h)
    return r


if __name__ == "__main__":
    sys.path[:0] = [os.path.join(ROOT, "ptrt-game-engine_amd"), os.path.join(ROOT, "oracle")]
    import oracle as O
    import ptrt_amd as P
    bn = P.blue_noise_table()
    for name in CASES:
        r = render_case(P, O, bn, name)
        np.savez_compressed(os.path.join(HERE, f"oracle_{name}.npz"), accum=r["accum"], depth=r["depth"],
                            object_id=r["object_id"], rgb8=r["rgb8"])
        print(name, r["stats"], r["accum"].mean(0))

#!/usr/bin/env python3
"""Soak check (GPU box, repo root): 24 consecutive 1080p frames with a wandering camera on five scenes, rendered
by the default traversal, by the plain one (batches of 64 pairs, every lane its own leaf, no stealing, no yield,
one lane per pair), by the lock-step mesh loop and (PMODE 1) by the lane-refill kernel; every frame's HDR image, object ids, generator states and ray
counts must be identical.  (tests/test_misc_gpu.py holds the two-frame version with the async and wavefront
kernels; this one is for more rays.)"""
import sys, os
sys.path.insert(0, "ptrt-game-engine_amd"); sys.path.insert(0, "tests")
import numpy as np, torch
import ptrt_amd as P
from test_parity_gpu import _many_meshes
def frames(build, opts, n, spp):
    s = P.Scene(1920, 1080); build(s)
    s.setPerfSamplesPerPixel(spp); s.setMaxBounceDepth(4); s.setDenoiserEnabled(False); s.setBloomEnabled(False)
    s.initBlueNoise(); s.uploadToGPU(); s.set_option("count_rays", 1)
    for k, v in opts.items(): s.set_option(k, v)
    out = []
    for f in range(n):
        if f % 5 == 4: s.moveCamera((0.3 * (f % 7) - 1.0, 0.2 * (f % 3), 5.0 - 0.1 * f))
        s.render_to_host()
        out.append((s.read(P.BUF_ACCUM).view(np.uint32).copy(), s.read(P.BUF_OBJECT_ID).copy(), s.read(P.BUF_RNG).copy(), s.stats()))
    s.close(); return out
plain = dict(fetch_min=0, leaf_pairs=0, steal=0, csteal=0, leaf_min=64, pair_split=0)
for name, build, spp in (("showcase", P.scenes.showcase, 4), ("fluid", lambda s: P.scenes.fluid(s, cells=256, t=0.7), 2),
                         ("many", lambda s: _many_meshes(P, s, n=60), 4), ("cornell", P.scenes.cornell, 4),
                         # duplicated / coplanar / degenerate triangles: first-found-wins decides, ~1 ray in 100 marked by a thief
                         ("coincident", lambda s: P.scenes.coincident(s, n=24, leaf=8), 4)):
    a = frames(build, {}, 24, spp); b = frames(build, plain, 24, spp); c = frames(build, dict(pair_trace=0), 24, spp)
    d = frames(build, dict(refill=2, persist=7), 24, spp)  # (PMODE 1: persistent waves with lane refill, 1,792 of them for 32,400 tiles)
    e = frames(build, dict(sample_sync=0), 24, spp)        # (every lane at its own pace; the default at 4 bounces keeps the samples in step)
    g = frames(build, dict(sample_sync=0, refill=2), 24, spp)
    # round 4: verified closest-hit subtree stealing (the default of PMODE 2 / 4; `a` samples both loop shapes on its frames 4-9) --
    # separate phases, merged loop, at its most eager without the thieves following their victims, and with every lane at its own pace
    more = [frames(build, o, 24, spp) for o in (dict(merged=0), dict(merged=1), dict(merged=1, csteal=1, csteal_leaf_min=4, csteal_follow=0),
                                                dict(merged=0, csteal=3, csteal_min=2, sample_sync=0))]
    bad = 0
    for f, (x, *others) in enumerate(zip(a, b, c, d, e, g, *more)):
        for k in range(3):
            if not all(np.array_equal(x[k], o[k]) for o in others): bad += 1
        if not all(x[3] == o[3] for o in others): bad += 1
    print(name, "24 frames x 10 variants:", "IDENTICAL" if bad == 0 else f"{bad} MISMATCHES", a[len(a) // 2][3], flush=True)

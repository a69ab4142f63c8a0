#!/usr/bin/env python3
"""Host + upload cost of moving ONE instance in the showcase scene (100,820 triangles + a 10,082-triangle instanced
sphere): Scene.commitObjectChanges() through ptrt_update_instances vs a full ptrt_upload_geometry."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ptrt-game-engine_amd"))
import torch  # noqa: E402,F401  (one HIP runtime per process: torch first)
import ptrt_amd as P  # noqa: E402

s = P.Scene(1920, 1080)
P.scenes.showcase(s)
ball = s.addSphere(71, P.Material((0.9, 0.2, 0.2), 0.3))
s.setPosition(ball, (0.0, 2.0, -5.0))
s.setPerfSamplesPerPixel(4)
s.setMaxBounceDepth(4)
s.setDenoiserEnabled(False)
s.setBloomEnabled(False)
s.initBlueNoise()
s.uploadToGPU()
s.render_to_host()
for label, touch in (("instance update", False), ("full upload", True)):
    t = []
    for k in range(10):
        s.setPosition(ball, (0.1 * k, 2.0, -5.0))
        if touch:
            s.scale(ball, (1.0001, 1.0, 1.0))     # a vertex change: forces BLAS rebuild + ptrt_upload_geometry
        t0 = time.perf_counter()
        s.commitObjectChanges()
        s.sync()
        t.append(time.perf_counter() - t0)
    t.sort()
    print(f"{label}: median {1e3 * t[len(t) // 2]:.3f} ms, min {1e3 * t[0]:.3f} ms")
s.close()

#!/usr/bin/env python3
"""A/B sweeps on the GPU box: one process, many (library variant is per process) option sets -> ms/frame table.
   python tools/sweep.py scene spp "opt=val,opt=val" "..." ...      (PTRT_AMD_LIB selects the library variant)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ptrt-game-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
torch.cuda.set_device(0)
import ptrt_amd as P  # noqa: E402
from bench import build_scene  # noqa: E402

scene, spp = sys.argv[1], int(sys.argv[2])
W, H = 1920, 1080
s = build_scene(P, scene, W, H, 0, 0, 0)
s.setPerfSamplesPerPixel(spp)
s.setMaxBounceDepth(4)
s.initBlueNoise()
s.uploadToGPU()
buf = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
defaults = dict(merged=0, lds_nodes=0, fetch_min=16, leaf_min=8, steal=1, leaf_pairs=1, pair_trace=1, stage=7, lds_pad=0, force_full=0)
specs = sys.argv[3:] or [""]
rounds = int(os.environ.get("SWEEP_ROUNDS", "3"))  # the list is walked `rounds` times (round-robin: drift hits every entry alike)
res = {spec: [] for spec in specs}
for _ in range(rounds):
    for spec in specs:
        opts = dict(defaults)
        for kv in filter(None, spec.split(",")):
            k, _, v = kv.partition("=")
            opts[k] = int(v)
        for k, v in opts.items():
            s.set_option(k, v)
        for _ in range(3):
            s.render_to_device(buf.data_ptr())
        torch.cuda.synchronize()
        n = 12
        for _ in range(n):
            s.render_to_device(buf.data_ptr())
        torch.cuda.synchronize()
        res[spec].append(float(s.kernel_ms_history(n).mean()))
for spec in specs:
    r = sorted(res[spec])
    print(f"{os.path.basename(os.environ.get('PTRT_AMD_LIB', 'default')):22s} {scene:9s} {spec or '(defaults)':40s} kernel ms: median {r[len(r) // 2]:7.3f}  min {r[0]:7.3f}  max {r[-1]:7.3f}", flush=True)

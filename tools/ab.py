#!/usr/bin/env python3
"""A/B of option sets on OVERLAPPING frames (two alternating device targets, as bench.py and a viewer render): wall time per
frame between synchronisations, plus the launches' own durations (ptrt_launch_ms_history) -- what tools/sweep.py, which renders
into one target (every frame ordered behind the stream) and reads the frame events, cannot see.
   python tools/ab.py <config | scene> [--frames N] [--rounds R] [--size WxH] [--spp S] [--depth D] "opt=val,opt=val" "..." ...
PTRT_AMD_LIB selects the library variant (per process)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ptrt-game-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
torch.cuda.set_device(0)
import ptrt_amd as P  # noqa: E402
from bench import CONFIGS, build_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("specs", nargs="*")
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--size", default=None)
ap.add_argument("--spp", type=int, default=None)
ap.add_argument("--depth", type=int, default=None)
ap.add_argument("--denoise", action="store_true")
ap.add_argument("--bloom", action="store_true")
ap.add_argument("--one-target", action="store_true", help="render into ONE target: no frame overlaps its predecessor")
a = ap.parse_intermixed_args()
cfg = dict(CONFIGS.get(a.config, dict(scene=a.config, width=1920, height=1080, spp=4, depth=4)))
if a.size:
    cfg["width"], cfg["height"] = (int(v) for v in a.size.split("x"))
if a.spp:
    cfg["spp"] = a.spp
if a.depth:
    cfg["depth"] = a.depth
W, H = cfg["width"], cfg["height"]
s = build_scene(P, cfg["scene"], W, H, 0, 0, 0)
s.setPerfSamplesPerPixel(cfg["spp"])
s.setMaxBounceDepth(cfg["depth"])
s.setDenoiserEnabled(a.denoise)
s.setBloomEnabled(a.bloom)
s.initBlueNoise()
s.uploadToGPU()
tgt = [torch.empty((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
specs = a.specs or [""]
base = {}
res = {spec: [] for spec in specs}
launch = {spec: [] for spec in specs}
info = {}
k = 0


def frame():
    global k
    s.render_to_device(tgt[0 if a.one_target else k & 1].data_ptr())
    k += 1


for _ in range(40):  # clocks, loop-shape choice
    frame()
torch.cuda.synchronize()
for r in range(a.rounds):
    for spec in specs:
        opts = {}
        for kv in filter(None, spec.split(",")):
            n, _, v = kv.partition("=")
            opts[n] = int(v)
        for n in opts:
            if n not in base:
                base[n] = s.get_option(n)
        for n, v in {**base, **opts}.items():
            s.set_option(n, v)
        s.set_option("time_launches", 1)
        for _ in range(8):
            frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.frames):
            frame()
        torch.cuda.synchronize()
        res[spec].append((time.perf_counter() - t0) / a.frames * 1e3)
        tr, tl = s.launch_ms_history(4 * a.frames)
        if len(tr):
            launch[spec].append((float(tr.mean()), float(tl.mean())))
        info[spec] = dict(pipelined=s.get_option("pipelined"), refilled=s.get_option("refilled"), pmode=s.get_option("pmode"),
                          merged=s.get_option("merged_eff"), split=s.get_option("split_eff"))
lib = os.path.basename(os.environ.get("PTRT_AMD_LIB", "default"))
for spec in specs:
    v = sorted(res[spec])
    ls = launch[spec]
    lt = f"launch {sum(x for x, _ in ls) / len(ls):.3f} ms + tail {sum(y for _, y in ls) / len(ls) * 1e3:.0f} us" if ls else "launch -"
    print(f"{lib:20s} {a.config:12s} {spec or '(defaults)':44s} ms/frame: median {v[len(v) // 2]:7.4f}  min {v[0]:7.4f}  max {v[-1]:7.4f}  {lt}  {info[spec]}",
          flush=True)
s.close()

#!/usr/bin/env python3
"""How many HIP streams may a process hold before two of the context's auxiliary streams share a hardware queue (and the two
launches of an overlapping frame run one after the other)?  Creates K extra streams (used once each), then times overlapping
frames.   python tools/stream_aliasing.py [config] [K ...]     env GPU_MAX_HW_QUEUES is the runtime's own knob (default 4)."""
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

late = "--late" in sys.argv  # create the extra streams AFTER the context has rendered overlapping frames (its streams exist and have run)
argv = [a for a in sys.argv if a != "--late"]
name = argv[1] if len(argv) > 1 else "showcase1080"
counts = [int(v) for v in argv[2:]] or [0, 1, 2, 3, 4, 6, 8]
cfg = CONFIGS[name]
W, H = cfg["width"], cfg["height"]
extra = []
def more(k):
    while len(extra) < k:
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            torch.zeros(16, device="cuda").add_(1)
        extra.append(st)
    torch.cuda.synchronize()


for k in counts:
    if not late:
        more(k)
    s = build_scene(P, cfg["scene"], W, H, 0, 0, 0)
    s.setPerfSamplesPerPixel(cfg["spp"])
    s.setMaxBounceDepth(cfg["depth"])
    s.initBlueNoise()
    s.uploadToGPU()
    tgt = [torch.empty((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    for f in range(40):
        s.render_to_device(tgt[f & 1].data_ptr())
    torch.cuda.synchronize()
    if late:
        more(k)
        for f in range(8):
            s.render_to_device(tgt[f & 1].data_ptr())
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 60
    for f in range(n):
        s.render_to_device(tgt[f & 1].data_ptr())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{'late ' if late else ''}GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(default)')}  {name}: {k} extra streams -> {ms:.4f} ms/frame "
          f"(pipelined {s.get_option('pipelined')}, split {s.get_option('split_eff')})", flush=True)
    s.close()

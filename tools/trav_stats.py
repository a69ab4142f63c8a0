#!/usr/bin/env python3
"""Lane-occupancy statistics of the PMODE 2 traversals (library built with -DPT_TRAV_STATS):
   PTRT_AMD_LIB=ptrt-game-engine_amd/build/variants/libptrt_stats.so python tools/trav_stats.py [scene] [W H spp]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ptrt-game-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
assert torch.cuda.is_available(), "needs a HIP device"
torch.cuda.set_device(0)
import ptrt_amd as P  # noqa: E402
from bench import build_scene  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "showcase"
W, H, spp = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1920, 1080, 4)
s = build_scene(P, scene, W, H, 0, 0, 0)
s.setPerfSamplesPerPixel(spp)
s.setMaxBounceDepth(4)
s.initBlueNoise()
s.uploadToGPU()
s.set_option("count_rays", 1)
for kv in sys.argv[5:]:
    k, _, v = kv.partition("=")
    s.set_option(k, int(v))
buf = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
out = (C.c_ulonglong * 32)()
P.lib.ptrt_debug_trav_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
P.lib.ptrt_debug_trav_stats(s.ctx, out)
s.stats()
s.render_to_device(buf.data_ptr())
torch.cuda.synchronize()
assert P.lib.ptrt_debug_trav_stats(s.ctx, out) == 0
outb = (C.c_ulonglong * 64)()
P.lib.ptrt_debug_trav_bounce.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert P.lib.ptrt_debug_trav_bounce(s.ctx, outb) == 0
st = s.stats()
v = list(out)
cy = dict(zip((8, 9, 10, 11, 12, 13, 14, 15), v[24:32]))  # cycle slots (CycleAcc)
print("rays", st)
if any(kv.startswith("async_lanes=1") for kv in sys.argv[5:]):
    it, sr, sl, mr, ml, nw, nl, _, tw, tl, tr = v[:11]
    print(f"async: main-loop iterations {it} ({it / (st['extension_rays'] + st['shadow_rays']) * 64:.1f} per 64 rays), "
          f"lanes tracing {100.0 * tr / max(1, it * 64):.1f} %")
    print(f"   shading block: {sr} runs ({100.0 * sr / it:.1f} % of iterations), lanes shading {100.0 * sl / max(1, sr * 64):.1f} %")
    print(f"   between-mesh block: {mr} runs ({100.0 * mr / it:.1f} % of iterations), lanes {100.0 * ml / max(1, mr * 64):.1f} %")
    print(f"   node steps: {nw} wave-iterations ({nw / it:.2f} per iteration), lanes busy {100.0 * nl / max(1, nw * 64):.1f} %")
    print(f"   triangle tests: {tw} wave-iterations ({tw / it:.2f} per iteration), lanes busy {100.0 * tl / max(1, tw * 64):.1f} %")
    sys.exit(0)
for name, b in (("closest", 0), ("any-hit", 8)):
    calls, pairs, nw, nl, lp, tw, tl, outer = v[b:b + 8]
    if not calls:
        continue
    print(f"{name}: calls {calls}  pairs/call {pairs / calls:.1f}  outer iters/call {outer / calls:.2f}")
    print(f"   node steps: {nw / calls:.1f} wave-iterations/call, lanes busy {100.0 * nl / max(1, nw * 64):.1f} %  "
          f"({nl / max(1, pairs):.1f} nodes per pair)")
    print(f"   leaf phases/call {lp / calls:.1f}; triangle loop: {tw / calls:.1f} wave-iterations/call, lanes busy "
          f"{100.0 * tl / max(1, tw * 64):.1f} %  ({tl / max(1, pairs):.1f} triangles per pair)")
vb = list(outb)
if hasattr(P.lib, "ptrt_debug_trav_dbg"):
    import struct
    dbg = (C.c_ulonglong * 1033)()
    P.lib.ptrt_debug_trav_dbg.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    P.lib.ptrt_debug_trav_dbg(s.ctx, dbg)
    ev = list(dbg[1025:1033])
    if any(ev):
        print(f"closest-hit stealing: {ev[0]} subtrees stolen; {ev[3]} rays traced again without it ({ev[1]} marks for a thief's hit in front of "
              f"its leaf box, {ev[2]} for equal distances from two walks of one pair)")
    if ev[4] or ev[5]:
        nw_c = v[2]
        print(f"   closest node loop, lanes per wave-iteration that take no step: {ev[4] / max(1, nw_c):.1f} wait at a leaf, "
              f"{ev[5] / max(1, nw_c):.1f} have no walk (separate phases' queue only)")
    if os.environ.get("PT_DBG"):
        f32 = lambda u: struct.unpack("f", struct.pack("I", u & 0xffffffff))[0]
        print("debug records:", dbg[1024])
        for i in range(min(40, dbg[1024])):
            o, k, w, t = dbg[4 * i:4 * i + 4]
            print(f"   old t={f32(o >> 32):.7g} order={(o >> 24) & 255} slot={o & 0xffffff}   key t={f32(k >> 32):.7g} order={(k >> 24) & 255} slot={k & 0xffffff}   "
                  f"ray {w & 255} lane {(w >> 8) & 255} thief {(w >> 16) & 1} nsteps {w >> 32} tcur {f32(t):.7g} bounce {t >> 32}")
if any(vb):  # the same counters by the rays' bounce (wave-uniform with the samples in step: sample_sync_eff)
    print(f"by bounce (samples in step: {s.get_option('sample_sync_eff')}):")
    for b in range(4):
        for name, o in (("closest", 0), ("any-hit", 8)):
            calls, pairs, nw, nl, lp, tw, tl, outer = vb[b * 16 + o:b * 16 + o + 8]
            if not calls:
                continue
            print(f"   bounce {b if b < 3 else '3+'} {name:8s}: calls {calls:8d}  pairs/call {pairs / calls:5.1f}  node loop {nw / calls:5.1f} wave-it/call at "
                  f"{100.0 * nl / max(1, nw * 64):4.1f} % busy ({nl / max(1, pairs):4.1f} nodes/pair)   triangle loop {tw / calls:5.1f} wave-it/call at "
                  f"{100.0 * tl / max(1, tw * 64):4.1f} % busy   share of all node wave-iterations {100.0 * nw / max(1, v[o + 2]):4.1f} %")
if cy[14]:  # instrumented build: where a wave's cycles go (s_memtime, summed over waves)
    tot = cy[14]
    pct = lambda k: 100.0 * cy[k] / tot
    print(f"wave cycles {tot:.4g}: closest-hit (or merged) trace {pct(12):.1f} %, shadow trace {pct(13):.1f} %, "
          f"everything else {100.0 - pct(12) - pct(13):.1f} %")
    if scene == "cornell" or os.environ.get("PT_TS_SHADING"):  # PMODE 1 has no queues: its build spends those slots on the shading
                                                               # phases; so does a -DPT_TS_SHADING build for every loop shape
        print(f"   shading: [A] regenerate {pct(8):.1f} %, [C] surface + light sample {pct(9):.1f} %, [C2] BSDF of the light sample "
              f"{pct(10):.1f} %, [E] scatter {pct(11):.1f} %, [R] refill {pct(15):.1f} %")
    else:
        print(f"   inside: queue runs {pct(8):.1f} %, closest node loops {pct(9):.1f} %, closest leaf blocks {pct(10):.1f} %, "
              f"any-hit queue runs {pct(11):.1f} %")
    if v[2] and v[5]:
        print(f"   per closest node wave-iteration {cy[9] / v[2]:.0f} units, per closest triangle wave-iteration {cy[10] / v[5]:.0f} units")
print(f"persistent loop: {v[16]} iterations, live lanes {100.0 * v[17] / max(1, v[16] * 64):.1f} %")
if v[16] and any(v[18:24]):  # who takes part in the phases of an iteration (lanes of 64, averaged over ALL iterations)
    it = v[16] * 64.0
    print(f"   of 64 lanes per iteration: [A] starts a sample {100 * v[18] / it:.1f} %, [C]/[E] shades a hit {100 * v[19] / it:.1f} %, [C] samples a "
          f"light {100 * v[20] / it:.1f} %, [D] walks its shadow ray {100 * v[21] / it:.1f} %" +
          (f", [R] takes a pixel {100 * v[22] / it:.1f} % (the block runs in {100.0 * v[23] / 64 / v[16]:.1f} % of the iterations)" if v[23] else ""))

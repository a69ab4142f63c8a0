#!/usr/bin/env python3
"""Every pair of fp32 significands through div3's core against the compiler's IEEE division (2^46 pairs, on the GPU box):
   python tools/div3_exhaustive.py [chunk]      -> one line per chunk of divisors, and a verdict"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ptrt-game-engine_amd"))
import ptrt_amd as P  # noqa: E402

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 3: the core on the RAW v_rcp_f32 (experiment)
s = P.Scene(16, 16)
P.lib.ptrt_debug_div3_check.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.POINTER(C.c_uint)]
bad = 0
t0 = time.time()
for first in range(0, 1 << 23, chunk):
    out = (C.c_uint * 9)()
    assert P.lib.ptrt_debug_div3_check(s.ctx, first, chunk, mode, out) == 0
    bad += out[0]
    print(f"divisors 1.m, m in [{first:#08x}, {first + chunk:#08x}) x 2^23 numerators: {out[0]} mismatches "
          f"{[hex(v) for v in list(out)[1:9]] if out[0] else ''}  ({time.time() - t0:.0f} s)", flush=True)
print(f"div3 core vs IEEE division over all 2^46 significand pairs: {bad} mismatches")
sys.exit(1 if bad else 0)

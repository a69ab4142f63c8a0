#!/usr/bin/env python3
"""Static instruction mix per phase of path_trace_kernel: build the ISA with -DPT_MARKS (PT_MARK leaves "; MARK x"
comments at the phase boundaries), then   python tools/asm_phases.py file.s <substring of the mangled kernel name>
Counts are STATIC (instructions in the text between two markers, in layout order) -- a guide to where the VALU
instructions of the loop body sit, not an execution profile."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and want in l]
for s in starts:
    e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
    print(lines[s].rstrip(":"))
    phase = "prologue"
    mix = collections.OrderedDict()
    for l in lines[s:e]:
        m = re.search(r"; MARK (\w+)", l)
        if m:
            phase = m.group(1)
            continue
        t = l.strip()
        if not l.startswith("\t") or t.startswith((".", ";")) or not t:
            continue
        op = t.split()[0]
        d = mix.setdefault(phase, collections.Counter())
        kind = ("valu" if op.startswith("v_") else "wait" if op.startswith("s_waitcnt") else "branch" if op.startswith(("s_cbranch", "s_branch"))
                else "smem" if op.startswith(("s_load", "s_buffer")) else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
                else "scratch" if op.startswith("scratch_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other")
        d[kind] += 1
        if op in ("v_div_scale_f32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32"):
            d[op] += 1
    for ph, d in mix.items():
        print(f"  {ph:9s}", "  ".join(f"{k} {v}" for k, v in d.items()))

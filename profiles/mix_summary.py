#!/usr/bin/env python3
"""gpurun_out/mix_<tag>/ (profiles/mix_pass.sh) -> profiles/<round>_<tag>_instruction_mix.json:  python profiles/mix_summary.py <tag> <round> "<bench args>" """

import collections, csv, glob, json, sys
tag, rnd, args = sys.argv[1], sys.argv[2], sys.argv[3]
agg, kernel = collections.defaultdict(list), None
for f in glob.glob(f"gpurun_out/mix_{tag}/p*/*/*_counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "path_trace" in r["Kernel_Name"]]
    if not rows:
        continue
    kernel = rows[-1]["Kernel_Name"]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if r["Kernel_Name"] == kernel})
    skip = set(ids[:5]) if len(ids) > 8 else set()
    for r in rows:
        if r["Kernel_Name"] == kernel and int(r["Dispatch_Id"]) not in skip:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
d = {}
if c.get("SQ_INSTS_VALU"):
    valu = c["SQ_INSTS_VALU"]
    d["active_lanes_per_valu_instruction"] = c.get("SQ_THREAD_CYCLES_VALU", 0) / valu
    arith = sum(c.get(k, 0) for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                                      "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT"))
    d["valu_arithmetic_share (fma+mul+add+int32+trans+cvt)"] = arith / valu
    d["branches_per_valu"] = c.get("SQ_INSTS_BRANCH", 0) / valu
    d["salu_cycles_per_valu"] = c.get("SQ_INST_CYCLES_SALU", 0) / valu
    d["useful_fp32_flop_per_launch (fma x2 + mul + add, x active lanes)"] = (
        (2 * c.get("SQ_INSTS_VALU_FMA_F32", 0) + c.get("SQ_INSTS_VALU_MUL_F32", 0) + c.get("SQ_INSTS_VALU_ADD_F32", 0)) * d["active_lanes_per_valu_instruction"])
json.dump({"what": f"rocprofv3 --pmc, bench.py {args} --no-pipeline; kernel {kernel}; mean per launch, first five launches dropped; three passes of 8 counters "
                   "(profiles/mix_pass.sh)", "kernel": kernel, "counters": c, "derived": d},
          open(f"profiles/{rnd}_{tag}_instruction_mix.json", "w"), indent=1)
print("instruction mix:", kernel, {k: round(v, 4) for k, v in d.items()})

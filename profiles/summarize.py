#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>/ (from profiles/pmc_pass.sh) into profiles/<round>_<tag>_{kernel_stats.csv,pmc.json}."""
import collections, csv, glob, json, os, shutil, sys
tag, rnd = sys.argv[1], sys.argv[2]
src = f"gpurun_out/prof_{tag}"
ks = sorted(glob.glob(f"{src}/stats/*/*_kernel_stats.csv"), key=os.path.getmtime)  # gpurun merges runs: newest wins
if ks:
    shutil.copy(ks[-1], f"profiles/{rnd}_{tag}_kernel_stats.csv")
out = {"tag": tag, "counters": {}}
newest = {}
for f in glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv"):
    d = os.path.dirname(os.path.dirname(f))
    if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
        newest[d] = f
for f in newest.values():
    agg = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(f)):
        if "path_trace" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Kernel_Name", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size") if k in r}
    for k, v in agg.items():
        out["counters"][k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    if meta:
        out["dispatch"] = meta
c = out["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    out["hbm_bytes_per_launch_uncorrected"] = (c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024
json.dump(out, open(f"profiles/{rnd}_{tag}_pmc.json", "w"), indent=1)
for k in sorted(c):
    print(f"{k:28s} {c[k]['mean_per_launch']:.4g}")
print(out.get("dispatch"))

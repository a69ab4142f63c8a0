#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>/ (from profiles/pmc_pass.sh) into profiles/<round>_<tag>_{kernel_stats.csv,pmc.json}."""
import collections, csv, glob, json, os, shutil, sys
tag, rnd = sys.argv[1], sys.argv[2]
src = f"gpurun_out/prof_{tag}"
DROP = int(os.environ.get("DROP", "15"))  # launches of the measured kernel left out: the bench's warm-up (pmc_pass.sh WARM)
ks = sorted(glob.glob(f"{src}/stats/*/*_kernel_stats.csv"), key=os.path.getmtime)  # gpurun merges runs: newest wins
if ks:  # rocprofv3's own table: every launch of the process, cold ones included
    shutil.copy(ks[-1], f"profiles/{rnd}_{tag}_kernel_stats_all_launches.csv")
out = {"tag": tag, "counters": {}}
# steady state from the per-dispatch trace of the same pass: per kernel, its launches in start order without the first DROP
# of the path-trace kernel the run settles on (and without every launch of a path-trace variant it only sampled)
kt = sorted(glob.glob(f"{src}/stats/*/*_kernel_trace.csv"), key=os.path.getmtime)
if kt:
    rows = sorted(csv.DictReader(open(kt[-1])), key=lambda r: int(r["Start_Timestamp"]))
    pt = [r for r in rows if "path_trace" in r["Kernel_Name"]]
    steady_name = pt[-1]["Kernel_Name"] if pt else None
    first_kept = None
    seen = 0
    per = collections.OrderedDict()
    for r in rows:
        name = r["Kernel_Name"]
        if "path_trace" in name:
            if name != steady_name:
                continue
            seen += 1
            if seen <= DROP:
                continue
            if first_kept is None:
                first_kept = int(r["Start_Timestamp"])
        elif first_kept is None:
            continue  # (launches before the first kept frame: set-up and warm-up)
        per.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in per.values()) or 1
    with open(f"profiles/{rnd}_{tag}_kernel_stats.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow([f"# steady state: rocprofv3 --kernel-trace of `bench.py` ({tag}), launches after the first {DROP} of the measured kernel; "
                    f"computed by profiles/summarize.py from the per-dispatch trace ({rnd}_{tag}_kernel_stats_all_launches.csv is rocprofv3's --stats table over every launch)"])
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            mean = sum(v) / len(v)
            sd = (sum((x - mean) ** 2 for x in v) / len(v)) ** 0.5
            w.writerow([name, len(v), sum(v), f"{mean:.1f}", f"{100.0 * sum(v) / total:.2f}", min(v), max(v), f"{sd:.1f}"])
    if steady_name and per.get(steady_name):
        v = per[steady_name]
        out["kernel"] = {"name": steady_name, "launches": len(v), "mean_ms": sum(v) / len(v) / 1e6, "min_ms": min(v) / 1e6, "max_ms": max(v) / 1e6,
                         "dropped_first": DROP}
        print("steady kernel:", out["kernel"])
newest = {}
for f in glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv"):
    d = os.path.dirname(os.path.dirname(f))
    if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
        newest[d] = f
for f in newest.values():
    agg = collections.defaultdict(list)
    meta = None
    rows = [r for r in csv.DictReader(open(f)) if "path_trace" in r["Kernel_Name"]]
    # (a scene whose loop shape is chosen by measurement launches two variants in its first frames: the one it settles on --
    # the last one dispatched -- is the one that counts)
    steady = rows[-1]["Kernel_Name"] if rows else None
    # (its first launches -- cold clocks -- are left out here as well)
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if r["Kernel_Name"] == steady}) if rows and "Dispatch_Id" in rows[0] else []
    skip = set(ids[:5]) if len(ids) > 8 else set()
    for r in rows:
        if r["Kernel_Name"] == steady and not (ids and int(r["Dispatch_Id"]) in skip):
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

# ---- inputs of bench.py's `roofline` block (profiles/<round>_roofline_inputs.json), keyed by bench --config name:
#   python profiles/summarize.py <tag> <round> <config>      e.g.  summarize.py cornell1080 r02 cornell1080
# VALU wave-instructions per launch (SQ_INSTS_VALU), HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (KB) with the
# gfx950 correction of MI355X_MICROARCH.md "HBM" (FETCH_SIZE tallies 128-B read requests at 64 B: doubled; other
# access widths are uncalibrated there, so the uncorrected sum is kept beside it), and the lane occupancy of the
# traversal loops from the instrumented build's report (profiles/<round>_lane_occupancy.txt, section "### <config>").
if len(sys.argv) > 3:
    import re
    cfg = sys.argv[3]
    path = f"profiles/{rnd}_roofline_inputs.json"
    try:
        allcfg = json.load(open(path))
    except Exception:
        allcfg = {}
    entry = {"source": f"profiles/{rnd}_{tag}_pmc.json"}
    if "SQ_INSTS_VALU" in c:
        entry["valu_wave_instructions_per_launch"] = round(c["SQ_INSTS_VALU"]["mean_per_launch"])
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f, w = c["FETCH_SIZE"]["mean_per_launch"] * 1024, c["WRITE_SIZE"]["mean_per_launch"] * 1024
        entry["hbm_bytes_per_launch"] = round(2 * f + w)
        entry["hbm_bytes_per_launch_uncorrected"] = round(f + w)
    occ = f"profiles/{rnd}_lane_occupancy.txt"
    if os.path.exists(occ):
        sect = re.split(r"^### ", open(occ).read(), flags=re.M)
        for s_ in sect:
            if s_.split("\n", 1)[0].strip() == cfg:
                lb = {}
                kind = None
                for line in s_.split("\n"):
                    m = re.match(r"(closest|any-hit):", line)
                    if m:
                        kind = m.group(1)
                    m = re.search(r"node steps: .*lanes busy ([0-9.]+) %", line)
                    if m and kind:
                        lb[f"{kind}_node_loop"] = round(float(m.group(1)) / 100, 4)
                    m = re.search(r"triangle loop: .*lanes busy ([0-9.]+) %", line)
                    if m and kind:
                        lb[f"{kind}_triangle_loop"] = round(float(m.group(1)) / 100, 4)
                    m = re.search(r"persistent loop: .*live lanes ([0-9.]+) %", line)
                    if m:
                        lb["live_lanes"] = round(float(m.group(1)) / 100, 4)
                entry["lane_busy"] = lb
    allcfg[cfg] = entry
    json.dump(allcfg, open(path, "w"), indent=1, sort_keys=True)
    print("roofline inputs:", cfg, entry)

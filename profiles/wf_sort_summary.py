#!/usr/bin/env python3
"""gpurun_out/wf_sort/s{0,1} (profiles/wf_sort_pass.sh) -> stdout: shade / trace kernel time per frame and the shade kernel's VALU lane
utilisation without and with active-path sorting."""
import collections, csv, glob, json, os
for S in (0, 1, 2, 4):
    if not os.path.exists(f"gpurun_out/wf_sort/s{S}/trace.log"):
        continue
    d = f"gpurun_out/wf_sort/s{S}"
    line = [l for l in open(f"{d}/trace.log") if l.startswith("{")]
    ms = json.loads(line[0])["ms_per_step"] if line else None
    kt = sorted(glob.glob(f"{d}/trace/*/*_kernel_trace.csv"), key=os.path.getmtime)
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[-1])):
        per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    frames = 15
    print(f"wf_sort={S}: bench ms/frame {ms}")
    for k, v in sorted(per.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
        if "wf_" in k:
            v = sorted(v)[len(v) // 3:]  # (steady state: the last two thirds of the launches)
            n_per_frame = len(per[k]) / frames
            print(f"   {k[:60]:60s} {len(per[k]):5d} launches ({n_per_frame:.1f} per frame), mean {sum(x[1] for x in v) / len(v) / 1e3:8.1f} us, per frame "
                  f"{sum(x[1] for x in v) / len(v) * n_per_frame / 1e6:6.3f} ms")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"{d}/pmc/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "wf_" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in agg.items():
        if c.get("SQ_ACTIVE_INST_VALU"):
            print(f"   {k[:60]:60s} VALU wave-instructions {c['SQ_INSTS_VALU']:.4g}, active lanes per VALU instruction "
                  f"{c['SQ_THREAD_CYCLES_VALU'] / c['SQ_INSTS_VALU']:.1f} of 64")

#!/usr/bin/env python3
"""Frames that overlap on the device (bench.py's default: ptrt_set_option "pipeline", and for PMODE 1 lane refill): what
rocprofv3 --kernel-trace saw.  A frame is `split` launches of the trace kernel (+ a tonemap pass each with lane refill) on
two auxiliary streams; launches of consecutive frames overlap, so a launch's duration is NOT a frame's: the frame interval is
the distance between the ends of consecutive frames' launches on the same stream.
   GPU box:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_overlap -- python3 $R/bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-configs3
   here:     python profiles/overlap_trace.py gpurun_out/prof_overlap > profiles/rNN_cornell1080_overlapped_frames.txt"""
import collections, csv, glob, os, sys
src = sys.argv[1]
kt = sorted(glob.glob(f"{src}/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
pt = [r for r in rows if "path_trace" in r["Kernel_Name"]]
name = collections.Counter(r["Kernel_Name"] for r in pt).most_common(1)[0][0]  # (the timed frames' kernel)
pt = [r for r in pt if r["Kernel_Name"] == name]
by_queue = collections.OrderedDict()
for r in pt:
    by_queue.setdefault(r.get("Queue_Id", "?"), []).append(r)
print(f"trace: {os.path.basename(kt)}")
print(f"kernel: {name}")
print(f"launches: {len(pt)} on {len(by_queue)} queue(s): " + ", ".join(f"queue {q}: {len(v)}" for q, v in by_queue.items()))
keep = 30  # the timed frames are the last ones of the run (the non-overlapped frames bench.py adds for kernel_ms come after them
           # and run a different instantiation when lane refill is on; with the same one they are cut off below by their gap)
for q, v in by_queue.items():
    v = v[-keep:] if len(v) > keep + 10 else v[len(v) // 2:]
    if len(v) < 3:
        continue
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in v]
    ends = [int(r["End_Timestamp"]) for r in v]
    iv = [(b - a) / 1e6 for a, b in zip(ends, ends[1:])]
    grid = v[-1].get("Grid_Size", "?")
    print(f"queue {q}: {len(v)} launches kept, grid {grid} threads: duration of a launch mean {sum(dur) / len(dur):.4f} ms "
          f"(min {min(dur):.4f}, max {max(dur):.4f}); FRAME INTERVAL (end to end of consecutive launches) mean {sum(iv) / len(iv):.4f} ms "
          f"(min {min(iv):.4f}, max {max(iv):.4f})")
others = collections.Counter()
odur = collections.defaultdict(float)
t0 = int(pt[len(pt) // 2]["Start_Timestamp"])
for r in rows:
    if int(r["Start_Timestamp"]) >= t0 and r["Kernel_Name"] != name:
        others[r["Kernel_Name"]] += 1
        odur[r["Kernel_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, n in others.most_common(6):
    print(f"other kernel since the middle of the run: {k}: {n} launches, mean {odur[k] / n * 1e3:.1f} us")

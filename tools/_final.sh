#!/bin/bash
# GPU box, repo root: the commands behind round 4's end-of-round tables.  Output under gpurun_out/final_r04/.
O=gpurun_out/final_r04; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/headline_driver_style.json 2> $O/err.txt            # Cornell 1080p + configs3 + cpu_baseline
python3 bench.py > $O/headline_40_100.json 2>> $O/err.txt
for c in showcase1080 showcase4k8 fluid; do python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $O/$c.json 2>> $O/err.txt; done
python3 bench.py --config fluid --via-commit --steps 20 --warmup 5 --no-cpu-baseline > $O/fluid_commit.json 2>> $O/err.txt
python3 bench.py --config fluid --rebuild --steps 20 --warmup 5 --no-cpu-baseline > $O/fluid_rebuild.json 2>> $O/err.txt
python3 bench.py --scene many --steps 20 --warmup 5 --no-cpu-baseline > $O/many.json 2>> $O/err.txt
for p in fast performance balanced quality ultra; do python3 bench.py --config million --preset $p --steps 6 --warmup 3 --no-cpu-baseline > $O/million_$p.json 2>> $O/err.txt; done
python3 bench.py --preset balanced --steps 20 --warmup 5 --no-cpu-baseline > $O/cornell_balanced.json 2>> $O/err.txt
python3 bench.py --preset balanced --opt atrous_exp=1 --steps 20 --warmup 5 --no-cpu-baseline > $O/cornell_balanced_fast_exp.json 2>> $O/err.txt
python3 bench.py --preset performance --steps 20 --warmup 5 --no-cpu-baseline > $O/cornell_performance.json 2>> $O/err.txt
python3 bench.py --present 2 --steps 20 --warmup 5 --no-cpu-baseline --no-configs3 > $O/cornell_present2.json 2>> $O/err.txt
python3 bench.py --farm 8 --steps 20 --warmup 5 --no-cpu-baseline --no-configs3 > $O/cornell_farm8.json 2>> $O/err.txt
# A/B of this round's options on overlapping frames (tools/ab.py): stealing on / off in both loop shapes
python tools/ab.py showcase1080 "merged=0,csteal=0" "merged=0" "merged=1,csteal=0" "merged=1" > $O/ab_showcase.txt 2>&1
python tools/ab.py cornell1080 "" "tm_prio=1" "tm_prio=2" "refill=0" > $O/ab_cornell.txt 2>&1
for f in $O/*.json; do python3 - "$f" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(f"{sys.argv[1].split('/')[-1]:36s} {d['ms_per_step']:9.4f} ms  {d['fps']:9.2f} fps  {d['value']:10.1f} Mrays/s  {d['config']['kernel'][:70]}", d.get("configs3", {}).get("ms_per_step", ""), d["config"].get("fluid_sources", ""), d["config"].get("present_ms_per_frame", ""), d["config"].get("farm_ms_per_frame", ""))
PY
done | tee $O/table.txt
grep -v amdgpu.ids $O/ab_showcase.txt $O/ab_cornell.txt

# GPU box, repo root: one line per workload -> stdout (the rows of DESIGN.md section 6).  --warmup 12 covers the frames on
# which a queue-mode scene samples its two loop shapes (bench.py adds them anyway: Farm.measure).
mkdir -p gpurun_out/r3x; cd gpurun_out/r3x; B="python ../../bench.py"
$B --steps 100 --warmup 40 > default.json 2> default.err
$B --config showcase1080 --steps 60 --warmup 40 > showcase1080.json 2>/dev/null
$B --config showcase4k8 --steps 8 --warmup 12 --no-cpu-baseline > showcase4k8.json 2>/dev/null
$B --config fluid --steps 100 --warmup 40 > fluid.json 2>/dev/null
$B --config fluid --steps 100 --warmup 40 --rebuild --no-cpu-baseline > fluid_rebuild.json 2>/dev/null
$B --scene many --steps 10 --warmup 4 --no-cpu-baseline --no-configs3 > many.json 2>/dev/null
$B --preset balanced --steps 100 --warmup 40 --no-cpu-baseline > cornell_balanced.json 2>/dev/null
$B --preset performance --steps 100 --warmup 40 --no-cpu-baseline > cornell_performance.json 2>/dev/null
$B --present 2 --steps 100 --warmup 40 --no-cpu-baseline --no-configs3 > present2.json 2>/dev/null
$B --farm 8 --steps 100 --warmup 40 --no-cpu-baseline --no-configs3 > farm8.json 2>/dev/null
$B --config showcase1080 --opt merged=1 --opt steal=0 --steps 60 --warmup 40 --no-cpu-baseline > showcase_merged.json 2>/dev/null
$B --config showcase1080 --opt merged=0 --steps 60 --warmup 40 --no-cpu-baseline > showcase_separate.json 2>/dev/null
$B --config showcase1080 --opt lds_nodes=1 --steps 60 --warmup 40 --no-cpu-baseline > showcase_ldsnodes.json 2>/dev/null
$B --config showcase1080 --opt wavefront=1 --steps 10 --warmup 4 --no-cpu-baseline > showcase_wavefront.json 2>/dev/null
for p in fast performance balanced quality; do $B --config million --preset $p --steps 60 --warmup 40 --no-cpu-baseline > million_$p.json 2>/dev/null; done
$B --config million --preset ultra --steps 2 --warmup 1 --no-cpu-baseline > million_ultra.json 2>/dev/null
for f in *.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    c=d["config"]; r=d["roofline"]
    print(f'{sys.argv[1]:28s} {d["ms_per_step"]:9.4f} ms  {d["fps"]:8.2f} fps  {d["value"]:10.1f} Mrays/s  kernel {r["kernel_ms"]}  valu {r["valu_issue_frac"]}  {c["workload"]} | {c["kernel"]}', c.get("present_ms_per_frame",""), c.get("farm_ms_per_frame",""), (d.get("configs3") or {}).get("ms_per_step",""), (d.get("cpu_baseline") or {}).get("value",""), ((d.get("cpu_baseline") or {}).get("single_thread") or {}).get("value",""), ((d.get("cpu_baseline") or {}).get("all_core") or {}).get("value",""))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done

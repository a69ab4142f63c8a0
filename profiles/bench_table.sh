#!/bin/bash
# GPU box, repo root: the rows of DESIGN.md section 6's tables -> gpurun_out/table.txt
O=gpurun_out/table.txt; : > $O
run() { echo "## $*" >> $O; python bench.py --no-cpu-baseline --no-configs3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('fps'), d['value'])" >> $O; }
run --config cornell1080
run --config showcase1080
run --config showcase4k8
run --config fluid
run --config fluid --rebuild
run --scene many
run --config cornell1080 --preset balanced
run --config cornell1080 --preset performance
run --config cornell1080 --present 2
run --config cornell1080 --farm 8
for p in fast performance balanced quality; do run --config million --preset $p; done
run --config million --preset ultra --steps 3 --warmup 1
cat $O

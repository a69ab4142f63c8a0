#!/bin/bash
# usage: profiles/pmc_pass.sh <tag> <bench args...>   (run on the GPU box from the repo root)
# Collects kernel stats and PMC counters in separate rocprofv3 passes (counters never combined
# with trace domains), writes gpurun_out/prof_<tag>/*.  The kernel-trace pass runs WARM untimed + 15 timed frames;
# profiles/summarize.py drops the first WARM launches of the measured kernel (cold clocks, and for the queue modes the
# frames on which the loop shape is chosen), so the committed average is the steady state bench.py times.  --no-pipeline:
# every frame is ONE launch ordered behind the stream, so a launch's duration is a frame's (the default overlaps frames).
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/stats -- python3 $R/bench.py --steps 15 --warmup ${WARM:-15} --no-pipeline --no-ramp --no-cpu-baseline --no-configs3 "$@" > $R/gpurun_out/prof_$TAG/stats.log 2>&1
# A queue-mode scene chooses between two shapes of its loop by measurement (ptrt_set_option "merged" = -1) and may choose differently
# from run to run: the counter passes are pinned to the shape the trace pass settled on, so that the file's durations and counters
# belong to ONE kernel.
PIN=$(python3 - "$R/gpurun_out/prof_$TAG/stats.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        c = json.loads(line)["config"]
        if c.get("pmode") in (2, 4):
            print(f"--opt merged={c['merged_eff']}")
PY
)
echo "loop shape pinned for the counter passes: ${PIN:-(none: the scene has one shape)}"
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_$N -- python3 $R/bench.py --steps 4 --warmup 12 --no-pipeline --no-ramp --no-cpu-baseline --no-configs3 "$@" $PIN > $R/gpurun_out/prof_$TAG/pmc_$N.log 2>&1 || echo "pass $N failed"
done

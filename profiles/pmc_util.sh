#!/bin/bash
# usage: profiles/pmc_util.sh <tag> <bench args...>   (GPU box, repo root)
# Lane utilisation and vector-memory pipe counters of the trace kernel, one rocprofv3 --pmc pass per group.
# (The TA_ADDR_STALLED_BY_TC/TA_DATA_STALLED_BY_TC/TA_ADDR_STALLED_BY_TD group hung rocprofv3 on this pool -- the run was
# killed after 7 silent minutes -- and is deliberately absent.)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/util_$TAG
i=0
for C in "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
         "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/util_$TAG/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/util_$TAG/p$i.log 2>&1 || echo "pass $i ($C) failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/util_$TAG/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "path_trace" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(f"{k:40s} {sum(agg[k])/len(agg[k]):.5g}")
PY

#!/bin/bash
# usage: profiles/wf_sort_pass.sh   (GPU box, repo root)  ->  profiles/r04_wavefront_sort.txt after profiles/wf_sort_summary.py
# Wavefront stages on the showcase 1080p frame without / with active-path sorting (option wf_sort): kernel times from one
# rocprofv3 --kernel-trace pass each, VALU lane utilisation of the shade kernel from one --pmc pass each (separate runs).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for S in ${WF_SORTS:-0 1 4}; do
  D=$R/gpurun_out/wf_sort/s$S; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/bench.py --config showcase1080 --steps 10 --warmup 5 --no-pipeline --no-ramp --no-cpu-baseline --no-configs3 --opt wavefront=1 --opt wf_sort=$S > $D/trace.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $D/pmc -- python3 $R/bench.py --config showcase1080 --steps 3 --warmup 2 --no-pipeline --no-ramp --no-cpu-baseline --no-configs3 --opt wavefront=1 --opt wf_sort=$S > $D/pmc.log 2>&1 || echo "pmc pass $S failed"
done

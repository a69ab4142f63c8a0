#!/bin/bash
# usage: profiles/mix_pass.sh <tag> <round> <bench args...>   (GPU box, repo root)
# Instruction mix of the path-trace kernel a bench run settles on: three rocprofv3 --pmc passes of 8 counters (never combined
# with a trace domain), every frame ONE launch ordered behind the stream (--no-pipeline) so that a dispatch is a frame.
# profiles/mix_summary.py (run again in the build container on the merged gpurun_out/mix_<tag>/) writes
# profiles/<round>_<tag>_instruction_mix.json (mean per launch over the launches after the first five).
TAG=$1; RND=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/mix_$TAG
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_IFETCH" \
         "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/mix_$TAG/p$i -- python3 $R/bench.py --steps 8 --warmup 8 --no-pipeline --no-ramp --no-cpu-baseline --no-configs3 "$@" > $R/gpurun_out/mix_$TAG/p$i.log 2>&1 || echo "mix pass $i failed"
done
cd $R
python3 profiles/mix_summary.py "$TAG" "$RND" "$*" || true

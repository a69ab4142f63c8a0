#!/bin/bash
# GPU box, repo root: the profile set of a round (RND, default r04) (kernel stats + PMC passes per config, lane occupancy of the instrumented
# build; `balanced` = Cornell with the denoiser + bloom chain).  Afterwards, in the build container:  for c in cornell1080 showcase1080 fluid many; do
#   python profiles/summarize.py $c r03 $c; done   and   cp gpurun_out/prof_lane/lane_occupancy.txt profiles/r03_lane_occupancy.txt
R=$PWD; V=ptrt-game-engine_amd/build/variants
CFGS=${1:-"cornell1080 cornell1080_refill showcase1080 fluid many balanced balanced_fast"}
mkdir -p gpurun_out/prof_lane
for c in $CFGS; do
  if [ $c = many ]; then bash profiles/pmc_pass.sh many --scene many; elif [ $c = balanced ]; then bash profiles/pmc_pass.sh balanced --preset balanced;
  elif [ $c = balanced_fast ]; then bash profiles/pmc_pass.sh balanced_fast --preset balanced --opt atrous_exp=1;
  elif [ $c = cornell1080_refill ]; then bash profiles/pmc_pass.sh cornell1080_refill --config cornell1080 --opt refill=2; bash profiles/mix_pass.sh cornell1080_refill ${RND:-r04} --config cornell1080 --opt refill=2;
  else bash profiles/pmc_pass.sh $c --config $c; fi
  echo "pmc $c done"
done
cd $R
( for c in "cornell1080:cornell 1920 1080 4" "showcase1080:showcase 1920 1080 4 merged=1" "fluid:fluid 1920 1080 2 merged=0" "many:many 1920 1080 4"; do echo "### ${c%%:*}"; PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py ${c#*:}; done
  # round 4: the showcase frame's closest-hit walks by bounce, without and with verified subtree stealing, in both loop shapes
  for o in "merged=0 csteal=0" "merged=0" "merged=1 csteal=0"; do echo "### showcase1080 $o (section above: merged=1 with stealing, the shipped loop)"; PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py showcase 1920 1080 4 $o; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_lane/lane_occupancy.txt
tail -5 gpurun_out/prof_lane/lane_occupancy.txt
# the default run's frames overlap (and PMODE 1 runs its lane-refill kernel): what the trace saw -> profiles/overlap_trace.py
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_overlap -- python3 $R/bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-configs3 > $R/gpurun_out/prof_overlap.log 2>&1 )
( echo "### cornell1080, lane refill (refill=2: every frame, also alone on the chip)"; PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py cornell 1920 1080 4 refill=2 ) 2>&1 | grep -v amdgpu.ids >> gpurun_out/prof_lane/lane_occupancy.txt
tail -3 gpurun_out/prof_lane/lane_occupancy.txt


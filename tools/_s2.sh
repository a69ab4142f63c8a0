#!/bin/bash
# round 4, GPU session 2: tests, per-bounce descent statistics, profile of the TIMED (lane-refill) headline kernel
O=gpurun_out/r4b; mkdir -p $O
V=ptrt-game-engine_amd/build/variants
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/tests.log
( for m in 0 1; do echo "### showcase1080 merged=$m"; PTRT_AMD_LIB=$V/libptrt_stats.so timeout -k 10 300 python tools/trav_stats.py showcase 1920 1080 4 merged=$m; done
  echo "### fluid"; PTRT_AMD_LIB=$V/libptrt_stats.so timeout -k 10 300 python tools/trav_stats.py fluid 1920 1080 2 merged=0
  echo "### million"; PTRT_AMD_LIB=$V/libptrt_stats.so timeout -k 10 300 python tools/trav_stats.py million 1920 1080 1 merged=0 ) 2>&1 | grep -v amdgpu.ids > $O/bounce_stats.txt
cat $O/bounce_stats.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs3 > $O/bench_5_20.json 2> $O/bench.err && echo bench ok
timeout -k 10 900 bash profiles/pmc_pass.sh cornell1080_refill --config cornell1080 --opt refill=2 && echo pmc ok
timeout -k 10 600 bash profiles/mix_pass.sh cornell1080_refill r04 --config cornell1080 --opt refill=2 && echo mix ok
R=$PWD; ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_overlap -- python3 $R/bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-configs3 > $R/gpurun_out/prof_overlap.log 2>&1 ) && echo overlap ok

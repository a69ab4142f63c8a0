#!/bin/bash
O=gpurun_out/r4c; mkdir -p $O
V=ptrt-game-engine_amd/build/variants
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "stealing or refill_thresholds" > $O/tests.log 2>&1; echo "tests rc=$?"
tail -5 $O/tests.log
timeout -k 10 400 python tools/ab.py showcase1080 "merged=0" "merged=0,csteal=1,csteal_min=0" "merged=0,csteal=1,csteal_min=8" "merged=0,csteal=2,csteal_min=8" "merged=0,csteal=2,csteal_min=16" "merged=0,csteal=4,csteal_min=12" "merged=0,csteal=1,csteal_min=24" "merged=1" > $O/ab_csteal.txt 2>&1
grep -v amdgpu.ids $O/ab_csteal.txt
timeout -k 10 300 python tools/ab.py showcase1080 --one-target "merged=0" "merged=0,csteal=1,csteal_min=8" "merged=0,csteal=2,csteal_min=16" > $O/ab_csteal_alone.txt 2>&1
grep -v amdgpu.ids $O/ab_csteal_alone.txt
( echo "### showcase1080 merged=0 csteal=1 csteal_min=8"; PTRT_AMD_LIB=$V/libptrt_stats.so timeout -k 10 300 python tools/trav_stats.py showcase 1920 1080 4 merged=0 csteal=1 csteal_min=8 ) 2>&1 | grep -v amdgpu.ids > $O/bounce_csteal.txt
cat $O/bounce_csteal.txt
timeout -k 10 300 python tools/ab.py fluid "merged=0" "merged=0,csteal=1,csteal_min=8" > $O/ab_fluid.txt 2>&1; grep -v amdgpu.ids $O/ab_fluid.txt
timeout -k 10 300 python tools/ab.py million --spp 1 "merged=0" "merged=0,csteal=1,csteal_min=8" > $O/ab_million.txt 2>&1; grep -v amdgpu.ids $O/ab_million.txt

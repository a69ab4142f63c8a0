#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
V=ptrt-game-engine_amd/build/variants
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "stealing or refill_thresholds" > $O/tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/tests.log
timeout -k 10 600 python tools/ab.py showcase1080 "merged=0" "merged=1" "merged=1,csteal=0" "merged=1,steal=1" "merged=1,steal=2" "merged=1,csteal=3" "merged=1,csteal_leaf_min=16" "merged=1,csteal_leaf_min=48" "merged=1,fetch_min=32" > $O/ab_merged.txt 2>&1; grep -v amdgpu.ids $O/ab_merged.txt
timeout -k 10 300 python tools/ab.py showcase1080 --one-target "merged=0" "merged=1" > $O/ab_alone.txt 2>&1; grep -v amdgpu.ids $O/ab_alone.txt
timeout -k 10 300 python tools/ab.py fluid "merged=0" "merged=1" > $O/ab_fluid.txt 2>&1; grep -v amdgpu.ids $O/ab_fluid.txt
timeout -k 10 300 python tools/ab.py million --spp 1 "merged=0" "merged=1" > $O/ab_million.txt 2>&1; grep -v amdgpu.ids $O/ab_million.txt
timeout -k 10 300 python tools/ab.py showcase4k8 --frames 8 --rounds 2 "merged=0" "merged=1" > $O/ab_4k.txt 2>&1; grep -v amdgpu.ids $O/ab_4k.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs3 > $O/bench_5_20.json 2> $O/bench.err; echo "bench rc=$?"

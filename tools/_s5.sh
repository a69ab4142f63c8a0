#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O
V=ptrt-game-engine_amd/build/variants
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "stealing" > $O/tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/tests.log
( for o in "csteal=1 csteal_min=0" "csteal=2 csteal_min=0" "csteal=2 csteal_min=0 csteal_follow=0"; do echo "### showcase merged=0 $o"; PTRT_AMD_LIB=$V/libptrt_stats.so timeout -k 10 300 python tools/trav_stats.py showcase 1920 1080 4 merged=0 $o; done ) 2>&1 | grep -v amdgpu.ids > $O/bounce_csteal.txt
grep "###\|stealing\|node steps\|triangle loop\|outer" $O/bounce_csteal.txt
timeout -k 10 600 python tools/ab.py showcase1080 "merged=0" "merged=0,csteal=1,csteal_min=0" "merged=0,csteal=2,csteal_min=0" "merged=0,csteal=3,csteal_min=0" "merged=0,csteal=4,csteal_min=0" "merged=0,csteal=2,csteal_min=4" "merged=0,csteal=2,csteal_min=8" "merged=0,csteal=2,csteal_min=0,csteal_follow=0" "merged=0,csteal=2,csteal_min=0,leaf_min=16" "merged=0,csteal=2,csteal_min=0,leaf_min=4" "merged=1" > $O/ab_csteal.txt 2>&1
grep -v amdgpu.ids $O/ab_csteal.txt
timeout -k 10 300 python tools/ab.py showcase1080 --one-target "merged=0" "merged=0,csteal=2,csteal_min=0" > $O/ab_csteal_alone.txt 2>&1
grep -v amdgpu.ids $O/ab_csteal_alone.txt
timeout -k 10 300 python tools/ab.py fluid "merged=0" "merged=0,csteal=2,csteal_min=0" "merged=0,csteal=2,csteal_min=8" > $O/ab_fluid.txt 2>&1; grep -v amdgpu.ids $O/ab_fluid.txt
timeout -k 10 300 python tools/ab.py million --spp 1 "merged=0" "merged=0,csteal=2,csteal_min=0" "merged=0,csteal=2,csteal_min=8" > $O/ab_million.txt 2>&1; grep -v amdgpu.ids $O/ab_million.txt
timeout -k 10 300 python tools/ab.py showcase4k8 --frames 8 --rounds 2 "merged=0" "merged=0,csteal=2,csteal_min=0" > $O/ab_4k.txt 2>&1; grep -v amdgpu.ids $O/ab_4k.txt

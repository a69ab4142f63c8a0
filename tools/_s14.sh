#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 600 python tools/ab.py showcase1080 "merged=1" "merged=1,sample_sync=0" "merged=1,sample_sync=0,fetch_min=32" "merged=1,sample_sync=0,csteal_leaf_min=16" "merged=0,sample_sync=0" > $O/ab_sync.txt 2>&1; grep -v amdgpu.ids $O/ab_sync.txt
timeout -k 10 300 python tools/ab.py showcase4k8 --frames 8 --rounds 2 "merged=1" "merged=1,sample_sync=0" > $O/ab_4k.txt 2>&1; grep -v amdgpu.ids $O/ab_4k.txt
timeout -k 10 300 python tools/ab.py fluid "merged=0" "merged=0,sample_sync=0" > $O/ab_fluid.txt 2>&1; grep -v amdgpu.ids $O/ab_fluid.txt
timeout -k 10 300 python tools/ab.py million --spp 1 "merged=0" "merged=0,sample_sync=0" > $O/ab_million.txt 2>&1; grep -v amdgpu.ids $O/ab_million.txt
timeout -k 10 300 python tools/ab.py million --spp 4 "merged=0" "merged=0,sample_sync=0" "merged=1" "merged=1,sample_sync=0" > $O/ab_million4.txt 2>&1; grep -v amdgpu.ids $O/ab_million4.txt

#!/bin/bash
O=gpurun_out/r4d; mkdir -p $O
V=ptrt-game-engine_amd/build/variants
( for o in "csteal=1 csteal_min=0"; do echo "### showcase merged=0 $o"; PT_DBG=1 PTRT_AMD_LIB=$V/libptrt_stats.so timeout -k 10 300 python tools/trav_stats.py showcase 1920 1080 4 merged=0 $o; done ) 2>&1 | grep -v amdgpu.ids > $O/dbg_csteal.txt
head -60 $O/dbg_csteal.txt

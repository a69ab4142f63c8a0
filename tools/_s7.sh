#!/bin/bash
O=gpurun_out/r4g; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"
tail -4 $O/tests.log
timeout -k 10 300 python -m pytest tests/test_denoiser.py -m gpu -q -s -k fast 2>&1 | grep "atrous_exp\|passed\|failed"
B="merged=0"
timeout -k 10 600 python tools/ab.py showcase1080 "merged=0,csteal=0" "$B" "$B,csteal_leaf_min=48" "$B,csteal_leaf_min=64" "$B,csteal_leaf_min=24" "$B,steal=2" "" > $O/ab_show.txt 2>&1; grep -v amdgpu.ids $O/ab_show.txt
timeout -k 10 300 python tools/ab.py cornell1080 --denoise --bloom "" "atrous_exp=1" > $O/ab_balanced.txt 2>&1; grep -v amdgpu.ids $O/ab_balanced.txt
timeout -k 10 300 python tools/ab.py cornell1080 --denoise --bloom --one-target "" "atrous_exp=1" > $O/ab_balanced1.txt 2>&1; grep -v amdgpu.ids $O/ab_balanced1.txt
bash profiles/pmc_pass.sh balanced --preset balanced > $O/pmc_balanced.log 2>&1; echo "pmc balanced done"
WARM=15 bash profiles/pmc_pass.sh balanced_fast --preset balanced --opt atrous_exp=1 > $O/pmc_balanced_fast.log 2>&1; echo "pmc balanced_fast done"
timeout -k 10 600 bash profiles/wf_sort_pass.sh > $O/wf_sort.log 2>&1; echo "wf_sort done"

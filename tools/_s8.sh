#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"
tail -4 $O/tests.log
WF_SORTS="0 1 2 4" timeout -k 10 900 bash profiles/wf_sort_pass.sh > $O/wf_sort.log 2>&1; echo "wf_sort done"
bash profiles/pmc_pass.sh balanced --preset balanced > $O/pmc_balanced.log 2>&1; echo "pmc balanced done"
bash profiles/pmc_pass.sh balanced_fast --preset balanced --opt atrous_exp=1 > $O/pmc_balanced_fast.log 2>&1; echo "pmc balanced_fast done"
timeout -k 10 300 python bench.py --config fluid --steps 40 --warmup 20 --no-cpu-baseline > $O/bench_fluid.json 2> $O/bench_fluid.err; echo "fluid rc=$?"
timeout -k 10 300 python tools/ab.py cornell1080 --denoise --bloom "" "atrous_exp=1" > $O/ab_balanced.txt 2>&1; grep -v amdgpu.ids $O/ab_balanced.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_5_20.json 2> $O/bench.err; echo "bench rc=$?"

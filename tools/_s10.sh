#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"
tail -4 $O/tests.log
timeout -k 10 600 python tools/ab.py showcase1080 "merged=1" "merged=1,steal=0" "" > $O/ab_merged.txt 2>&1; grep -v amdgpu.ids $O/ab_merged.txt
timeout -k 10 300 python tools/ab.py showcase4k8 --frames 8 --rounds 2 "merged=1" "merged=1,steal=0" > $O/ab_4k.txt 2>&1; grep -v amdgpu.ids $O/ab_4k.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_5_20.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --config showcase1080 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_show.json 2>> $O/bench.err; echo "bench show rc=$?"
timeout -k 10 300 python bench.py --config fluid --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_fluid.json 2>> $O/bench.err; echo "bench fluid rc=$?"
timeout -k 10 300 python bench.py --config million --preset ultra --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_ultra.json 2>> $O/bench.err; echo "bench ultra rc=$?"
timeout -k 10 300 python bench.py --scene many --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_many.json 2>> $O/bench.err; echo "bench many rc=$?"

#!/bin/bash
O=gpurun_out/r4k; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_bench_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
RND=r04 timeout -k 10 1500 bash profiles/collect.sh > $O/collect.log 2>&1; echo "collect rc=$?"; tail -5 $O/collect.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_5_20.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --config showcase1080 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_show.json 2>> $O/bench.err; echo "bench show rc=$?"

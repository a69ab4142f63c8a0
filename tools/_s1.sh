#!/bin/bash
# round 4, GPU session 1: tests, headline bench three ways, tonemap-priority A/B
O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_5_20.json 2> $O/bench_5_20.err && echo bench1 ok
timeout -k 10 300 python bench.py --steps 100 --warmup 40 --no-cpu-baseline --no-configs3 > $O/bench_40_100.json 2>> $O/bench_5_20.err && echo bench2 ok
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs3 --no-time-launches > $O/bench_5_20_notl.json 2>> $O/bench_5_20.err && echo bench3 ok
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs3 --no-ramp > $O/bench_5_20_noramp.json 2>> $O/bench_5_20.err && echo bench4 ok
timeout -k 10 300 python tools/ab.py cornell1080 "" "tm_prio=1" "tm_prio=2" "tm_prio=3" "refill=0" > $O/ab_tm.txt 2>&1
cat $O/ab_tm.txt | grep -v amdgpu.ids
timeout -k 10 300 python tools/ab.py showcase1080 "" "merged=0" "merged=1" > $O/ab_show.txt 2>&1
grep -v amdgpu.ids $O/ab_show.txt

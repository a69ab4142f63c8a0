#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2>> $O/bench.err; echo "bench default rc=$?"

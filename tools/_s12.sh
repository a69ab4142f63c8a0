#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
PTRT_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 4 --width 640 --height 360 --steps 3 --warmup 1 > $O/rehearse4.json 2> $O/rehearse4.err; echo "rehearse4 rc=$?"
PTRT_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 3 --width 641 --height 357 --steps 3 --warmup 1 --layout bands > $O/rehearse3.json 2> $O/rehearse3.err; echo "rehearse3 rc=$?"
timeout -k 10 900 python tools/soak_variants.py > $O/soak.txt 2>&1; echo "soak rc=$?"; grep -v amdgpu.ids $O/soak.txt

mkdir -p gpurun_out/r2r
python -m pytest tests/test_config3_gpu.py tests/test_presets_gpu.py -m gpu -x -q -s > gpurun_out/r2r/tests.log 2>&1; tail -5 gpurun_out/r2r/tests.log; grep "rays per band" gpurun_out/r2r/tests.log
python bench.py --steps 30 --warmup 3 > gpurun_out/r2r/bench_default.json 2> gpurun_out/r2r/bench_default.err; tail -c 2500 gpurun_out/r2r/bench_default.json
for c in showcase1080 showcase4k8 fluid; do python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2r/bench_$c.json 2>&1; done
for p in fast performance balanced quality; do python bench.py --config million --preset $p --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2r/bench_million_$p.json 2>&1; done
python bench.py --config million --preset ultra --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2r/bench_million_ultra.json 2>&1
for f in gpurun_out/r2r/bench_*.json; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"workload": "[^"]*"' $f | head -1; done

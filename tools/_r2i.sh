mkdir -p gpurun_out/r2i
python -m pytest tests/test_parity_gpu.py tests/test_instances_gpu.py tests/test_refit.py -m gpu -x -q > gpurun_out/r2i/tests.log 2>&1; tail -3 gpurun_out/r2i/tests.log
( python tools/sweep.py many 4 "" pair_trace=0
  python tools/sweep.py showcase 4 "" merged=0 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2i/out.txt

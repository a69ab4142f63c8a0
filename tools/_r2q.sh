mkdir -p gpurun_out/r2q
python -m pytest tests/test_parity_gpu.py tests/test_wavefront_gpu.py tests/test_misc_gpu.py -m gpu -x -q > gpurun_out/r2q/tests.log 2>&1; tail -3 gpurun_out/r2q/tests.log
( python tools/sweep.py showcase 4 "" merged=1,steal=0
  python tools/sweep.py fluid 2 ""
  python tools/sweep.py cornell 4 ""
  python tools/sweep.py many 4 "" ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2q/out.txt

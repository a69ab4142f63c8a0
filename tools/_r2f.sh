mkdir -p gpurun_out/r2f
python -m pytest tests/test_parity_gpu.py tests/test_wavefront_gpu.py tests/test_post.py tests/test_denoiser.py -m gpu -x -q > gpurun_out/r2f/tests.log 2>&1; tail -3 gpurun_out/r2f/tests.log
( python tools/sweep.py showcase 4 "" merged=0 steal=0 steal=4 "steal=0,fetch_min=8" "steal=0,leaf_min=4" "steal=0,leaf_min=16"
  python tools/sweep.py fluid 2 "" merged=0 steal=0
  python tools/sweep.py cornell 4 ""
  python tools/sweep.py many 4 "" ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2f/sweep.txt

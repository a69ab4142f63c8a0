mkdir -p gpurun_out/r2h; V=ptrt-game-engine_amd/build/variants
python -m pytest tests/test_parity_gpu.py tests/test_wavefront_gpu.py tests/test_instances_gpu.py tests/test_refit.py -m gpu -x -q > gpurun_out/r2h/tests.log 2>&1; tail -3 gpurun_out/r2h/tests.log
( python tools/sweep.py showcase 4 "" merged=0 steal=0
  python tools/sweep.py fluid 2 "" merged=0 steal=0
  python tools/sweep.py cornell 4 ""
  PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py showcase 1920 1080 4 steal=0
  ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2h/out.txt

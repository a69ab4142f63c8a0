mkdir -p gpurun_out/r2m; V=ptrt-game-engine_amd/build/variants
python -m pytest tests/test_parity_gpu.py tests/test_instances_gpu.py tests/test_refit.py tests/test_misc_gpu.py -m gpu -x -q > gpurun_out/r2m/tests.log 2>&1; tail -3 gpurun_out/r2m/tests.log
( python tools/sweep.py many 4 ""
  PTRT_AMD_LIB=$V/libptrt_slots4.so python tools/sweep.py many 4 ""
  PTRT_AMD_LIB=$V/libptrt_slots1.so python tools/sweep.py many 4 ""
  python tools/sweep.py showcase 4 "" merged=0 steal=0
  python tools/sweep.py cornell 4 ""
  python tools/sweep.py fluid 2 "" merged=0
  PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py many 1920 1080 4 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2m/out.txt

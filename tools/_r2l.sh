mkdir -p gpurun_out/r2l; V=ptrt-game-engine_amd/build/variants
python -m pytest tests/test_parity_gpu.py tests/test_instances_gpu.py -m gpu -x -q > gpurun_out/r2l/tests.log 2>&1; tail -2 gpurun_out/r2l/tests.log
( python tools/sweep.py many 4 ""

  PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py many 1920 1080 4 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2l/out.txt

mkdir -p gpurun_out/r2k; V=ptrt-game-engine_amd/build/variants
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "tlas or many" > gpurun_out/r2k/tests.log 2>&1; tail -2 gpurun_out/r2k/tests.log
( python tools/sweep.py many 4 ""
  PTRT_AMD_LIB=$V/libptrt_chunk8.so python tools/sweep.py many 4 ""
  PTRT_AMD_LIB=$V/libptrt_chunk2.so python tools/sweep.py many 4 "" ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2k/out.txt

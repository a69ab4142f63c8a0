mkdir -p gpurun_out/r2j; V=ptrt-game-engine_amd/build/variants
( PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py many 1920 1080 4
  PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py showcase 1920 1080 4 merged=0 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2j/out.txt

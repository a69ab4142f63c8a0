mkdir -p gpurun_out/r2g; V=ptrt-game-engine_amd/build/variants
( PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py showcase 1920 1080 4 steal=0
  PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py fluid 1920 1080 2 steal=0
  python tools/sweep.py showcase 4 "" merged=0 steal=0
  python tools/sweep.py fluid 2 "" merged=0 steal=0
  python tools/sweep.py cornell 4 "" ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2g/out.txt

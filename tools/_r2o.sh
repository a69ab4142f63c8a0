mkdir -p gpurun_out/r2o; V=ptrt-game-engine_amd/build/variants
( PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py showcase 1920 1080 4
  PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py fluid 1920 1080 2
  python tools/sweep.py showcase 4 "" steal=0 steal=2 leaf_min=4 leaf_min=16 fetch_min=8 fetch_min=32 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2o/out.txt

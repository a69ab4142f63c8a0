mkdir -p gpurun_out/r2d; V=ptrt-game-engine_amd/build/variants
SPECS='"" merged=0 fetch_min=8 fetch_min=32 fetch_min=64 leaf_min=4 leaf_min=16 leaf_min=32 steal=2 steal=4 steal=0 fetch_min=32,leaf_min=16 fetch_min=8,leaf_min=4,steal=2'
( eval python tools/sweep.py showcase 4 $SPECS
  eval python tools/sweep.py fluid 2 '""' merged=0 fetch_min=32 leaf_min=16
  for v in w3 w5; do PTRT_AMD_LIB=$V/libptrt_$v.so python tools/sweep.py showcase 4 "" merged=0; PTRT_AMD_LIB=$V/libptrt_$v.so python tools/sweep.py fluid 2 "" merged=0; PTRT_AMD_LIB=$V/libptrt_$v.so python tools/sweep.py cornell 4 ""; done
  python tools/sweep.py cornell 4 ""
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2d/sweep.txt

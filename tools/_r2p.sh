mkdir -p gpurun_out/r2p
( python tools/sweep.py showcase 4 "" steal=0 steal=2 steal=4 leaf_min=4 leaf_min=12 leaf_min=16 fetch_min=8 fetch_min=24 fetch_min=32 "leaf_min=12,steal=2" merged=1,steal=0
  python tools/sweep.py fluid 2 "" leaf_min=4 leaf_min=16 fetch_min=8 fetch_min=32 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2p/out.txt

set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hostile_geometry_gpu.py -m gpu -q > gpurun_out/hostile.log 2>&1
export PTRT_AMD_LIB=ptrt-game-engine_amd/build/variants/libptrt_stats.so
for sc in coincident coincident_tlas; do
 for o in "merged=0 csteal=2" "merged=1 csteal=2"; do
  echo "### $sc $o" >> gpurun_out/coin_stats.txt
  timeout -k 10 200 python3 tools/trav_stats.py $sc 1920 1080 4 $o >> gpurun_out/coin_stats.txt 2>&1
 done
done

set -e
cd $GRAFT_REPO_ROOT
for cfg in showcase1080 fluid; do
  timeout -k 10 200 python3 tools/ab.py $cfg "sample_sync=1" "sample_sync=0" --frames 40 --rounds 3 >> gpurun_out/ab_ss.txt 2>&1
done
timeout -k 10 200 python3 tools/ab.py showcase4k8 "sample_sync=1" "sample_sync=0" --frames 8 --rounds 3 >> gpurun_out/ab_ss.txt 2>&1
timeout -k 10 200 python3 tools/ab.py many "sample_sync=1" "sample_sync=0" --frames 8 --rounds 3 >> gpurun_out/ab_ss.txt 2>&1

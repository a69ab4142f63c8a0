set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "tiny_and_ragged or showcase_small" > gpurun_out/lpt_tests.log 2>&1 || { tail -30 gpurun_out/lpt_tests.log; exit 1; }
o=gpurun_out/ab_tile_run.txt; rm -f $o
S="tile_run=0 tile_run=8 tile_run=516 tile_run=520 tile_run=1028 tile_run=1032 tile_run=2056 tile_run=528"
for cfg in showcase1080 fluid; do
  timeout -k 10 300 python3 tools/ab.py $cfg $S --frames 30 --rounds 3 --one-target >> $o 2>&1
  timeout -k 10 300 python3 tools/ab.py $cfg $S --frames 30 --rounds 3 >> $o 2>&1
done
timeout -k 10 300 python3 tools/ab.py showcase4k8 $S --frames 8 --rounds 3 >> $o 2>&1
for v in 8 1028; do
  timeout -k 10 300 python3 bench.py --config fluid --opt tile_run=$v --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fluid tile_run=$v', d['ms_per_step'])" >> $o
done

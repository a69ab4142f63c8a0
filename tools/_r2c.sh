mkdir -p gpurun_out/r2c && cd gpurun_out/r2c && R=../..
python $R/bench.py --scene showcase --steps 20 --warmup 3 --no-cpu-baseline > showcase.json 2>&1
python $R/bench.py --scene showcase --steps 20 --warmup 3 --no-cpu-baseline --opt merged=0 > showcase_m0.json 2>&1
python $R/bench.py --scene fluid --spp 2 --steps 20 --warmup 3 --no-cpu-baseline > fluid.json 2>&1
python $R/bench.py --scene fluid --spp 2 --steps 20 --warmup 3 --no-cpu-baseline --opt merged=0 > fluid_m0.json 2>&1
PTRT_AMD_LIB=$R/ptrt-game-engine_amd/build/variants/libptrt_stats.so python $R/tools/trav_stats.py showcase 1920 1080 4 > stats_merged.txt 2>&1
PTRT_AMD_LIB=$R/ptrt-game-engine_amd/build/variants/libptrt_stats.so python $R/tools/trav_stats.py showcase 1920 1080 4 merged=0 > stats_m0.txt 2>&1
for f in *.json; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f; done; cat stats_merged.txt stats_m0.txt

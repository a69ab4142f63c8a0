R=$PWD; V=ptrt-game-engine_amd/build/variants
mkdir -p gpurun_out/r2w
for c in cornell1080 showcase1080 fluid; do bash profiles/pmc_pass.sh $c --config $c; echo "pmc $c done"; done
bash profiles/pmc_pass.sh many --scene many; echo "pmc many done"
cd $R
( for c in "showcase1080:showcase 1920 1080 4" "fluid:fluid 1920 1080 2" "many:many 1920 1080 4"; do echo "### ${c%%:*}"; PTRT_AMD_LIB=$V/libptrt_stats.so python tools/trav_stats.py ${c#*:}; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r2w/lane_occupancy.txt
tail -5 gpurun_out/r2w/lane_occupancy.txt

mkdir -p gpurun_out/r2s
python -m pytest tests/test_farm_gpu.py tests/test_tilefarm_post.py tests/test_present_gpu.py tests/test_capi_symbols.py -x -q > gpurun_out/r2s/tests.log 2>&1; tail -4 gpurun_out/r2s/tests.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs3 --farm 8 > gpurun_out/r2s/farm8.json 2>gpurun_out/r2s/farm8.err; tail -c 600 gpurun_out/r2s/farm8.json; tail -3 gpurun_out/r2s/farm8.err
python bench.py --config showcase4k8 --steps 6 --warmup 2 --no-cpu-baseline --farm 8 > gpurun_out/r2s/farm8_4k.json 2>&1; tail -c 400 gpurun_out/r2s/farm8_4k.json
PTRT_BENCH_REHEARSE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r2s/rehearse2.json 2>gpurun_out/r2s/rehearse2.err; tail -c 900 gpurun_out/r2s/rehearse2.json; tail -3 gpurun_out/r2s/rehearse2.err
python - <<'PY'
import sys; sys.path.insert(0,'ptrt-game-engine_amd')
import torch, ptrt_amd as P
W,H=3840,2160
per=[]
for r in range(8):
    s=P.Scene(W,H,interleave=(r,8)); P.scenes.showcase(s); s.setPerfSamplesPerPixel(8); s.setMaxBounceDepth(4); s.setDenoiserEnabled(False); s.setBloomEnabled(False); s.initBlueNoise(); s.uploadToGPU(); s.set_option("count_rays",1)
    s.render_to_host(); st=s.stats(); per.append(st["extension_rays"]+st["shadow_rays"]); s.close()
print("rays per strip context:", per, "max/mean", max(per)/(sum(per)/8))
PY

import sys, time
sys.path.insert(0, "ptrt-game-engine_amd")
import torch
import ptrt_amd as P
W, H = 1920, 1080
for parts in (1, 2):
    for barrier in (0, 1):
        tf = P.TileFarm(W, H, [0] * parts, strips=True)
        for sc in tf.scenes:
            P.scenes.cornell(sc); sc.setPerfSamplesPerPixel(4); sc.setMaxBounceDepth(4)
            sc.setDenoiserEnabled(False); sc.setBloomEnabled(False); sc.initBlueNoise(); sc.uploadToGPU()
        tgt = [torch.empty((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
        for i in range(12):
            tf.render_to_device(tgt[i & 1].data_ptr())
        tf.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            tf.render_to_device(tgt[i & 1].data_ptr())
            if barrier:
                torch.cuda.synchronize()
        tf.sync(); torch.cuda.synchronize()
        print(f"parts {parts} barrier-between-frames {barrier}: {(time.perf_counter() - t0) / 40 * 1e3:.4f} ms/frame")
        tf.close()

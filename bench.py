#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X path-tracing render loop.

One "step" = one frame of the hot path (acceleration-structure check, path trace + fused
tonemap, and for N > 1 the RCCL gather of the tile images to rank 0).  At N = 1 the workload is
BASELINE.json configs[1]: Cornell box, 1920x1080, 4 spp, 4 bounces.  For N > 1 the SAME frame is
split into N horizontal bands (one process per GPU, scene replicated, per-pixel random streams
keyed by the global pixel index, so the image is bit-identical to the 1-GPU frame) -> strong
scaling.  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 30 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ptrt-game-engine_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
ALGO_BYTES_PER_PIXEL = 143.0   # RNG 48R+48W, accum 12W, normal 12W, depth 4W, id 4W, tonemap 12R+3W


def build_scene(P, name, W, H, y0, rows, device):
    s = P.Scene(W, H, tile_y0=y0, tile_rows=rows, device=device)
    if name == "cornell":
        P.scenes.cornell(s)
    elif name == "showcase":
        P.scenes.showcase(s)
    elif name == "many":  # 128 spheres/cubes of 2 x 32^2 triangles behind a real TLAS (not a BASELINE config)
        P.scenes.many(s, 128, sphere_segments=32)
    elif name == "fluid":
        s.water_mesh, _ = P.scenes.fluid(s, cells=256, t=0.0)
    else:
        raise SystemExit(f"unknown scene {name}")
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    return s


def cpu_baseline(P, scene_name, W, H, spp, depth, frame, threads):
    """The oracle ("port" of the reference path; the reference has no CPU renderer) timed on this
    box's host cores over a bounded sample: the full frame once for Cornell."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    s = build_scene(P, scene_name, W, H, 0, 0, P.HOST_ONLY)
    s.setPerfSamplesPerPixel(spp)
    s.setMaxBounceDepth(depth)
    desc = s.flatten()
    rows = H if scene_name == "cornell" else max(8, H // 4)
    y0 = (H - rows) // 2
    rng = O.xorwow_init(P.DEFAULT_SEED, y0 * W, rows * W)
    bn = P.blue_noise_table()
    t0 = time.perf_counter()
    r = O.render(desc, W, H, spp, depth, frame, bn, rng, tile_y0=y0, tile_rows=rows, threads=threads)
    dt = time.perf_counter() - t0
    rays = r["stats"]["extension_rays"] + r["stats"]["shadow_rays"]
    s.close()
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"rows {y0}..{y0 + rows} of one {W}x{H} frame, {spp} spp, {depth} bounces, "
                      f"{rays} rays in {dt:.2f} s", "fps_equivalent": round(rows / H / dt, 4) if rows else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--denoise", action="store_true",
                    help="also run motion vectors + the spatiotemporal denoiser each frame (N=1 only; not the headline)")
    ap.add_argument("--bloom", action="store_true", help="also run the bloom chain (N=1 only; not the headline)")
    ap.add_argument("--scale", type=float, default=1.0,
                    help="perfSettings.resolutionScale: trace at scale*size, bilinear up-scale (N=1 only; not the headline)")
    ap.add_argument("--present", type=int, default=0, metavar="SLOTS",
                    help="N=1 only, not the headline: after the timed loop also run the viewer loop (device frame -> pinned "
                         "host, SLOTS-deep ring) and report its PCIe-inclusive ms/frame as config.present_ms_per_frame")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="ptrt_set_option for A/B experiments (e.g. pair_trace=0); not for the headline")
    ap.add_argument("--rebuild", action="store_true",
                    help="fluid scene: rebuild the water BVH on the GPU every frame (ptrt_build_bvh) instead of refitting it")
    args = ap.parse_args()

    import torch
    import ptrt_amd as P

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the render loop has no CPU path")
    # PTRT_BENCH_REHEARSE=1: dry run of the N > 1 code path on a ONE-GPU box -- every rank renders its band on
    # cuda:0 and the gather goes through gloo with host staging.  Not a measurement (the JSON line says so).
    rehearse = os.environ.get("PTRT_BENCH_REHEARSE") == "1" and world > 1
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from ptrt_amd import tilefarm
    W, H = args.width, args.height
    y0, rows = tilefarm.bands(H, world)[rank]  # horizontal bands; the last rank takes the remainder rows
    scene = build_scene(P, args.scene, W, H, y0 if world > 1 else 0, rows if world > 1 else 0, dev_index)
    scene.setPerfSamplesPerPixel(args.spp)
    scene.setMaxBounceDepth(args.depth)
    post_on_rank0 = world > 1 and (args.denoise or args.bloom)
    if world == 1:
        scene.setDenoiserEnabled(args.denoise)
        scene.setBloomEnabled(args.bloom)
        scene.setResolutionScale(args.scale)
    scene.initBlueNoise()
    scene.uploadToGPU()
    scene.set_option("count_rays", 1)
    for kv in args.opt:
        name, _, value = kv.partition("=")
        scene.set_option(name, int(value))
    # render on torch's current stream so the RCCL gather is ordered after the frame without host syncs
    stream = torch.cuda.current_stream()
    scene.set_stream(stream.cuda_stream)

    # two band images + two assembled frames: frame i's gather (RCCL's own stream) overlaps frame
    # i+1's render; a buffer is reused only after its gather has completed
    tiles = [torch.empty((rows, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    views = [None, None]
    if world > 1 and rank == 0:
        frames = [torch.empty((H, W, 3), dtype=torch.uint8, device="cpu" if rehearse else "cuda") for _ in range(2)]
        views = [tilefarm.frame_views(f, H, world) for f in frames]
    pending = [None, None]

    # --denoise / --bloom with N > 1 (not the headline): the bands send HDR + G-buffers (32 B/px) instead of RGB8 and
    # rank 0 runs the post chain over the gathered frame in a second, full-frame context (tilefarm.gather_gbuffers,
    # Scene.post_frame -> ptrt_post_frame)
    presenter = gviews = gframe = gband = out_frame = None
    if post_on_rank0:
        kinds = dict(accum=P.BUF_ACCUM, normal=P.BUF_NORMAL, depth=P.BUF_DEPTH, object_id=P.BUF_OBJECT_ID)
        gband = {k: torch.as_tensor(scene.device_array(kind), device="cuda") for k, kind in kinds.items()}
        if rehearse:
            gband_host = None
        if rank == 0:
            presenter = build_scene(P, args.scene, W, H, 0, 0, dev_index)
            presenter.setPerfSamplesPerPixel(args.spp)
            presenter.setMaxBounceDepth(args.depth)
            presenter.setDenoiserEnabled(args.denoise)
            presenter.setBloomEnabled(args.bloom)
            presenter.initBlueNoise()
            presenter.uploadToGPU()
            presenter.set_stream(stream.cuda_stream)
            dev = "cpu" if rehearse else "cuda"
            gframe = {k: torch.empty((H * W, c), dtype=getattr(torch, dt), device=dev) for k, c, dt in tilefarm.GBUFFER_KINDS}
            gviews = tilefarm.gbuffer_views(gframe, H, world)
            out_frame = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")

    # config 5 ("fluid"): every step first moves the water surface (new vertex positions already in
    # HBM), refits the BVH on the GPU, then traces: the refit+trace pipeline, no host sync inside
    water = None
    if args.scene == "fluid":
        import numpy as np
        water = [torch.from_numpy(np.ascontiguousarray(P.scenes.water_vertices(256, t / 60.0))).cuda()
                 for t in range(8)]
    counter = [0, 0]

    def step():
        if water is not None:
            move = scene.rebuildFromDevice if args.rebuild else scene.refitFromDevice
            move(scene.water_mesh, water[counter[0] % len(water)].data_ptr())
            counter[0] += 1
        b = counter[1] & 1
        counter[1] += 1
        if pending[b] is not None:
            pending[b].wait()
        scene.render_to_device(tiles[b].data_ptr())
        if post_on_rank0:
            src = {k: (t.cpu() if rehearse else t) for k, t in gband.items()}
            tilefarm.gather_gbuffers(dist, src, gviews, rank, world, H)
            if rank == 0:
                fr = {k: (t.cuda() if rehearse else t) for k, t in gframe.items()}
                presenter.post_frame(fr["accum"].data_ptr(), fr["normal"].data_ptr(), fr["depth"].data_ptr(),
                                     fr["object_id"].data_ptr(), out_frame.data_ptr())
            return
        if rehearse:
            tilefarm.gather_bands(dist, tiles[b].cpu(), views[b], rank, world, H)
        else:
            pending[b] = tilefarm.gather_bands(dist, tiles[b], views[b], rank, world, H, async_op=True)

    def fence():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    scene.stats()  # reset counters
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0

    st = scene.stats()
    rays = float(st["extension_rays"] + st["shadow_rays"])
    kms = scene.kernel_ms_history(args.steps)
    kernel_ms = float(kms.mean()) if len(kms) else float("nan")
    if world > 1:
        t = torch.tensor([dt, rays, kernel_ms], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        rays = float(tsum[1])
        kernel_ms = float(tmax[2])
    if rank != 0:
        dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    mrays = rays / dt / 1e6
    fps = args.steps / dt
    tile_pixels = W * rows  # rank 0's launch
    algo_bytes = ALGO_BYTES_PER_PIXEL * tile_pixels
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms == kernel_ms and kernel_ms > 0 else None
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if world == 1 and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"{args.scene}_{W}x{H}_{args.spp}spp_{args.depth}b"
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        **({"rehearsal": "gloo + host staging on one GPU: NOT a measurement"} if rehearse else {}),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "fps": round(fps, 2),
        "rays_per_frame": round(rays / args.steps),
        "config": {"workload": f"{args.scene} {W}x{H} {args.spp}spp {args.depth}-bounce"
                               + (" +denoise" if args.denoise else "") + (" +bloom" if args.bloom else "")
                               + (f" scale{args.scale}" if args.scale != 1.0 else "")
                               + (" +gpu-rebuild" if args.rebuild else ""), "scene": args.scene,
                   "width": W, "height": H, "spp": args.spp, "max_depth": args.depth,
                   "parallelism": (f"tile{world}" + ("+post-on-rank0" if post_on_rank0 else "")) if world > 1 else "single",
                   "kernel": "path_trace_kernel (megakernel, fused tonemap)"},
        "roofline": {"bound": "hbm", "achieved": None if achieved is None else round(achieved, 3),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 6),
                     "traffic": traffic, "kernel_ms": round(kernel_ms, 4),
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "note": "path is VALU/latency bound, not HBM bound (SURVEY 8(d)); see DESIGN.md"},
    }
    if args.present > 0 and world == 1:
        scene.set_stream(0)  # the viewer loop runs on the context's own stream
        scene.view_run(args.warmup + 2, slots=args.present, keep=False)
        _, pms = scene.view_run(args.steps, slots=args.present, keep=False)
        out["config"]["present_ms_per_frame"] = round(pms, 4)
        out["config"]["present_slots"] = args.present
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(P, args.scene, W, H, args.spp, args.depth, 0,
                                           min(16, len(os.sched_getaffinity(0))))
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X path-tracing render loop.

One "step" = one frame of the hot path (acceleration-structure check, path trace + fused
tonemap, and for N > 1 the RCCL gather of the tile images to rank 0).  At N = 1 the workload is
BASELINE.json configs[1]: Cornell box, 1920x1080, 4 spp, 4 bounces.  For N > 1 the SAME frame is
split into N horizontal bands (one process per GPU, scene replicated, per-pixel random streams
keyed by the global pixel index, so the image is bit-identical to the 1-GPU frame) -> strong
scaling.  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 30 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
         --master-port P bench.py --gpus N --steps K --warmup W

`--config` picks another BASELINE configuration as the measured workload (the headline stays configs[1]):
  cornell1080   configs[1]  Cornell box 1920x1080, 4 spp, 4 bounces                (default)
  showcase1080  configs[2]  showcase scene (100,820 triangles) 1920x1080, 4 spp
  showcase4k8   configs[3]  showcase scene 3840x2160, 8 spp (the frame the 8-GPU split is quoted on)
  fluid         configs[4]  water refit + trace, 1920x1080, 2 spp
  million                   8 x 125,000-triangle spheres: the workload of the reference's published fps
                            (Test game screenshots/readme.txt), with --preset fast|performance|balanced|quality|ultra
Every default run also measures configs[3] over the same N ranks for a few frames and reports it in the extra field
"configs3" of the line, so a scaling run over N = 1, 2, 4, 8 yields BASELINE's curve without changing the headline.
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
SIMDS, CLOCK_HZ, VALU_CYCLES = 256 * 4, 2.4e9, 2.0  # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, a wave64 VALU op issues over 2 cycles

CONFIGS = {
    "cornell1080": dict(scene="cornell", width=1920, height=1080, spp=4, depth=4),
    "showcase1080": dict(scene="showcase", width=1920, height=1080, spp=4, depth=4),
    "showcase4k8": dict(scene="showcase", width=3840, height=2160, spp=8, depth=4),
    "fluid": dict(scene="fluid", width=1920, height=1080, spp=2, depth=4),
    "million": dict(scene="million", width=1920, height=1080, spp=1, depth=4),
}


def build_scene(P, name, W, H, y0, rows, device, interleave=None):
    s = P.Scene(W, H, tile_y0=y0, tile_rows=rows, device=device, interleave=interleave)
    if name == "cornell":
        P.scenes.cornell(s)
    elif name == "showcase":
        P.scenes.showcase(s)
    elif name == "many":  # 128 spheres/cubes of 2 x 32^2 triangles behind a real TLAS (not a BASELINE config)
        P.scenes.many(s, 128, sphere_segments=32)
    elif name == "fluid":
        s.water_mesh, _ = P.scenes.fluid(s, cells=256, t=0.0)
    elif name == "million":
        P.scenes.million(s)
    elif name in ("coincident", "coincident_tlas"):  # duplicated / coplanar / degenerate triangles (tests/test_hostile_geometry_gpu.py)
        P.scenes.coincident(s, n=24, leaf=8 if name == "coincident" else 2)
    elif name == "matrix":  # the reference application's scene 10 (app_utils.cuh:729-795)
        P.scenes.material_matrix(s)
    else:
        raise SystemExit(f"unknown scene {name}")
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    return s


def cpu_quota():
    """CPU time this process's cgroup may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None if unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(p), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / p, 2)
    except Exception:
        return None


def cpu_baseline(P, scene_name, W, H, spp, depth, frame, threads):
    """The oracle ("port" of the reference path; the reference has no CPU renderer) timed on this box's host cores
    over a bounded sample of the same frame: all-core on a band of rows, and one core on a narrower band."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    s = build_scene(P, scene_name, W, H, 0, 0, P.HOST_ONLY)
    s.setPerfSamplesPerPixel(spp)
    s.setMaxBounceDepth(depth)
    desc = s.flatten()
    bn = P.blue_noise_table()

    def sample(rows, nthreads):
        y0 = (H - rows) // 2
        rng = O.xorwow_init(P.DEFAULT_SEED, y0 * W, rows * W)
        t0 = time.perf_counter()
        r = O.render(desc, W, H, spp, depth, frame, bn, rng, tile_y0=y0, tile_rows=rows, threads=nthreads)
        dt = time.perf_counter() - t0
        rays = r["stats"]["extension_rays"] + r["stats"]["shadow_rays"]
        return rays, dt, y0

    rows = H if scene_name == "cornell" else max(8, H // 4)
    rays, dt, y0 = sample(rows, threads)
    rows1 = max(8, rows // 16)
    rays1, dt1, y01 = sample(rows1, 1)
    # every hardware thread this process may use (1-core AND all-core); a second frame's worth of rows
    # when the first took under two seconds, so that thread start-up is not what is timed
    allc = len(os.sched_getaffinity(0))
    all_core = None
    if allc > threads:
        raysA, dtA, y0A = sample(rows, allc)
        all_core = {"value": round(raysA / dtA / 1e6, 3), "unit": "Mrays/s", "cores": allc,
                    "sample": f"rows {y0A}..{y0A + rows}, {raysA} rays in {dtA:.2f} s",
                    # (a container's CPU quota caps what `cores` threads can use: when it is below them this is the
                    # quota's worth of cores, not the machine's)
                    "cgroup_cpu_quota_cores": cpu_quota()}
    s.close()
    out = {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
           "host_cores": os.cpu_count(),
           "sample": f"rows {y0}..{y0 + rows} of one {W}x{H} frame, {spp} spp, {depth} bounces, "
                     f"{rays} rays in {dt:.2f} s", "fps_equivalent": round(rows / H / dt, 4) if rows else None,
           "single_thread": {"value": round(rays1 / dt1 / 1e6, 3), "unit": "Mrays/s", "cores": 1,
                             "sample": f"rows {y01}..{y01 + rows1}, {rays1} rays in {dt1:.2f} s"}}
    if all_core:
        out["all_core"] = all_core
    return out


class Farm:
    """One workload on this rank: its band of the frame, double-buffered band images and (rank 0) assembled frames."""

    def __init__(self, P, torch, dist, tilefarm, env, scene_name, W, H, spp, depth, args=None, preset=None):
        self.P, self.torch, self.dist, self.tilefarm, self.env = P, torch, dist, tilefarm, env
        rank, world, dev, rehearse = env["rank"], env["world"], env["dev"], env["rehearse"]
        self.W, self.H, self.scene_name = W, H, scene_name
        self.denoise = bool(args and args.denoise)
        self.bloom = bool(args and args.bloom)
        # N > 1: interleaved 8-row strips (every rank samples the whole frame: the showcase frame's worst BAND carries
        # 1.6x the mean rays, three of eight being sky) unless --layout bands; the post chain on rank 0 gathers bands
        self.strips = world > 1 and not (self.denoise or self.bloom) and (args is None or args.layout == "strips")
        self.y0, self.rows = tilefarm.bands(H, world)[rank]  # horizontal bands; the last rank takes the remainder rows
        if self.strips:
            s = self.scene = build_scene(P, scene_name, W, H, 0, 0, dev, interleave=(rank, world))
            self.rows = s.tile_rows
        else:
            s = self.scene = build_scene(P, scene_name, W, H, self.y0 if world > 1 else 0, self.rows if world > 1 else 0, dev)
        s.setPerfSamplesPerPixel(spp)
        s.setMaxBounceDepth(depth)
        self.rebuild = bool(args and args.rebuild)
        self.post_on_rank0 = world > 1 and (self.denoise or self.bloom)
        if world == 1 and args is not None:
            s.setDenoiserEnabled(args.denoise)
            s.setBloomEnabled(args.bloom)
            s.setResolutionScale(args.scale)
            if preset:  # Scene::setPerformancePreset (scene.cuh:1833-1879): depth, scale, denoiser, bloom; ultra also 128 spp
                s.setPerformancePreset(preset)
        s.initBlueNoise()
        s.uploadToGPU()
        s.set_option("count_rays", 1)
        for kv in (args.opt if args else []):
            name, _, value = kv.partition("=")
            s.set_option(name, int(value))
        if args is not None and args.no_pipeline:
            s.set_option("pipeline", 0)
        # render on torch's current stream so the RCCL gather is ordered after the frame without host syncs
        self.stream = torch.cuda.current_stream()
        s.set_stream(self.stream.cuda_stream)
        # two band images + two assembled frames: frame i's gather (RCCL's own stream) overlaps frame
        # i+1's render; a buffer is reused only after its gather has completed
        trows = tilefarm.max_strip_rows(H, world) if self.strips else self.rows  # (strip images are padded to one size)
        self.tiles = [torch.empty((trows, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
        self.views = [None, None]
        self.frames = self.frames_ext = self.parts = self.assemblers = None
        if world > 1 and rank == 0:
            fdev = "cpu" if rehearse else "cuda"
            # (one spare row behind the frame: where the strip layout's padding rows go, tilefarm.StripAssembler)
            self.frames_ext = [torch.empty((H + 1, W, 3), dtype=torch.uint8, device=fdev) for _ in range(2)]
            self.frames = [f[:H] for f in self.frames_ext]
            self.views = [tilefarm.frame_views(f, H, world) for f in self.frames]
            if self.strips:
                self.assemblers = [tilefarm.StripAssembler(torch, H, W, world, fdev) for _ in range(2)]
                self.parts = [a.parts for a in self.assemblers]
        self.pending = [None, None]
        # --denoise / --bloom with N > 1 (not the headline): the bands send HDR + G-buffers (32 B/px) instead of RGB8
        # and rank 0 runs the post chain over the gathered frame in a second, full-frame context
        # (tilefarm.gather_gbuffers, Scene.post_frame -> ptrt_post_frame)
        self.presenter = self.gviews = self.gframe = self.gband = self.out_frame = None
        if self.post_on_rank0:
            kinds = dict(accum=P.BUF_ACCUM, normal=P.BUF_NORMAL, depth=P.BUF_DEPTH, object_id=P.BUF_OBJECT_ID)
            self.gband = {k: torch.as_tensor(s.device_array(kind), device="cuda") for k, kind in kinds.items()}
            if rank == 0:
                pr = self.presenter = build_scene(P, scene_name, W, H, 0, 0, dev)
                pr.setPerfSamplesPerPixel(spp)
                pr.setMaxBounceDepth(depth)
                pr.setDenoiserEnabled(self.denoise)
                pr.setBloomEnabled(self.bloom)
                pr.initBlueNoise()
                pr.uploadToGPU()
                pr.set_stream(self.stream.cuda_stream)
                d = "cpu" if rehearse else "cuda"
                self.gframe = {k: torch.empty((H * W, c), dtype=getattr(torch, dt), device=d) for k, c, dt in tilefarm.GBUFFER_KINDS}
                self.gviews = tilefarm.gbuffer_views(self.gframe, H, world)
                self.out_frame = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
        # config 5 ("fluid"): every step first moves the water surface (new vertex positions already in
        # HBM), refits the BVH on the GPU, then traces: the refit+trace pipeline, no host sync inside
        # --fluid-source: where the new positions come from.  "device": already in HBM (a simulation on the GPU);
        # "host": pinned host memory, so every step also crosses PCIe (4.7 MB, SURVEY's "upload verts -> refit -> trace");
        # "commit": the reference's own caller -- what updatePTScene does to a `Triangles` mesh (vertices and faces rewritten on
        # the host, dirty flags set) followed by the unchanged commitObjectChanges(), under the GpuRefit policy
        self.water = self.water_host = None
        self.fluid_source = (args.fluid_source if args else "device")
        if scene_name == "fluid":
            import numpy as np
            host = [np.ascontiguousarray(P.scenes.water_vertices(256, t / 60.0)) for t in range(8)]
            self.water = [torch.from_numpy(a).cuda() for a in host]
            self.water_host = [torch.from_numpy(a).pin_memory() for a in host]
            self.water_np = host
            s.setDynamicGeometryPolicy("GpuRebuild" if self.rebuild else "GpuRefit")  # (only "commit" goes through it)
        self.counter = [0, 0]

    def step(self):
        s, env, tf = self.scene, self.env, self.tilefarm
        rank, world, rehearse = env["rank"], env["world"], env["rehearse"]
        if self.water is not None:
            i = self.counter[0] % len(self.water)
            if self.fluid_source == "commit":
                s.setTriangleSoup(s.water_mesh, self.water_np[i])
                s.commitObjectChanges()
            elif self.fluid_source == "host" and not self.rebuild:
                s.refitFromHost(s.water_mesh, self.water_host[i].data_ptr())
            else:
                move = s.rebuildFromDevice if self.rebuild else s.refitFromDevice
                move(s.water_mesh, self.water[i].data_ptr())
            self.counter[0] += 1
        b = self.counter[1] & 1
        self.counter[1] += 1
        self._retire(b)
        if self.pending[b ^ 1] is not None and s.get_option("pipelined"):
            # Frames overlap (ptrt_set_option "pipeline"): the NEXT frame, into the other buffer, will wait only for what is on
            # the scene's stream when THIS render call is made -- so that buffer's consumer, the collective of the previous
            # frame (it runs on the communicator's stream), has to be on the scene's stream by now (include/ptrt.h).  The
            # wait is the stream's, not the host's, and this frame's launches do not wait for the stream.
            self._retire(b ^ 1)
        s.render_to_device(self.tiles[b].data_ptr())
        if self.post_on_rank0:
            src = {k: (t.cpu() if rehearse else t) for k, t in self.gband.items()}
            tf.gather_gbuffers(self.dist, src, self.gviews, rank, world, self.H)
            if rank == 0:
                fr = {k: (t.cuda() if rehearse else t) for k, t in self.gframe.items()}
                self.presenter.post_frame(fr["accum"].data_ptr(), fr["normal"].data_ptr(), fr["depth"].data_ptr(),
                                          fr["object_id"].data_ptr(), self.out_frame.data_ptr())
            return
        if self.strips:
            parts = self.parts[b] if rank == 0 else None
            if rehearse:
                tf.gather_strips(self.dist, self.tiles[b].cpu(), None, parts, None, rank, world, async_op=True).wait()
                if rank == 0:
                    self.assemblers[b].scatter(self.frames_ext[b])
            else:
                self.pending[b] = tf.gather_strips(self.dist, self.tiles[b], None, parts, None, rank, world, async_op=True)
        elif rehearse:
            tf.gather_bands(self.dist, self.tiles[b].cpu(), self.views[b], rank, world, self.H)
        else:
            self.pending[b] = tf.gather_bands(self.dist, self.tiles[b], self.views[b], rank, world, self.H, async_op=True)

    def _retire(self, b):
        """Completes frame buffer b's gather (and, for strips, scatters the parts into the frame on rank 0)."""
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
            if self.strips and self.env["rank"] == 0:
                self.assemblers[b].scatter(self.frames_ext[b])

    def fence(self):
        for b in (0, 1):
            self._retire(b)
        if self.env["world"] > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def _max_over_ranks(self, x):
        if self.env["world"] == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.env["rehearse"] else "cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def ramp(self, block=10, cap=300, tol=0.01, budget_s=2.0):
        """Clock-ramp guard: untimed frames, in blocks of `block`, until THREE consecutive blocks' mean frame times agree within
        `tol` (a fresh box runs its first frames at lower clocks and the ramp takes tens of milliseconds: 5 + 20 frames used to
        read 5 % slower than 40 + 100).  At most `cap` frames or `budget_s` seconds; every rank takes the same decision (the MAX
        over ranks is compared)."""
        n, hist, t_all = 0, [], time.perf_counter()
        self.fence()
        while n < cap:
            t0 = time.perf_counter()
            for _ in range(block):
                self.step()
            self.fence()
            hist.append(self._max_over_ranks((time.perf_counter() - t0) / block))
            n += block
            done = len(hist) >= 3 and max(hist[-3:]) - min(hist[-3:]) <= tol * min(hist[-3:])
            over = self._max_over_ranks(time.perf_counter() - t_all) > budget_s
            if done or over:
                break
        self.ramp_ms = [round(h * 1e3, 4) for h in hist]
        return n

    def measure(self, steps, warmup, ramp=True, time_launches=True):
        """W untimed frames (+ the frames on which a scene settles its loop shape, + the clock-ramp guard's), then exactly
        `steps` frames between barrier + synchronize brackets; the MAX over ranks."""
        torch, env = self.torch, self.env
        for _ in range(warmup):
            self.step()
        # a queue-mode scene samples both shapes of its loop on its first frames (ptrt_set_option "merged" = -1: frames 4-9,
        # decision at frame 10, one host wait): those frames stay outside the timed region whatever --warmup says
        self.tuning_frames = 0
        while self.tuning_frames < 16 and not self.scene.get_option("merged_decided"):
            self.step()
            self.tuning_frames += 1
        # events around every launch of the timed frames, on the stream the launch runs on (ptrt_launch_ms_history): overlapping
        # frames are two launches each on auxiliary streams, and what the roofline is priced on is THOSE launches.  (Switched on
        # in front of the ramp frames, so that the events exist before the timed region starts.)
        self.scene.set_option("time_launches", 1 if time_launches else 0)
        self.ramp_ms = []
        self.ramp_frames = self.ramp() if ramp else 0
        if not ramp and time_launches:
            for _ in range(2):
                self.step()
        self.fence()
        self.scene.stats()  # reset counters
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        dt = time.perf_counter() - t0
        st = self.scene.stats()
        rays = float(st["extension_rays"] + st["shadow_rays_walked"])  # rays actually traced
        rays_ref = float(st["extension_rays"] + st["shadow_rays"])     # rays the reference path traces
        overlapped = bool(self.scene.get_option("pipelined"))
        refilled = bool(self.scene.get_option("refilled"))  # (PMODE 1: persistent waves whose lanes draw the next pixel)
        split = int(self.scene.get_option("split_eff"))
        kms = self.scene.kernel_ms_history(steps)
        kernel_ms = float(kms.mean()) if len(kms) else float("nan")
        launch = None
        tr, tl = self.scene.launch_ms_history(steps * 4)
        self.scene.set_option("time_launches", 0)
        if overlapped and len(tr):
            # the timed frames' own launches: duration of the trace kernel of each, and of what follows it on its stream (the
            # tonemap pass of the lane-refill kernel, including its wait for places on the chip)
            launch = dict(n=int(len(tr)), per_frame=split, ms=float(tr.mean()), ms_min=float(tr.min()), ms_max=float(tr.max()),
                          tail_ms=float(tl.mean()))
            kernel_ms = launch["ms"]
        kernel_alone_ms = None
        if overlapped:
            # The timed frames overlapped on the device (ptrt_set_option "pipeline": a frame's launches follow the previous
            # frame's and do not wait for it to drain), so the events around a frame measured the frame INTERVAL.  For
            # reference, beside the timed launches' own durations: a few more frames, untimed, each ONE launch ordered behind
            # the stream -- a frame alone on the chip, which is what the rocprofv3 passes with --no-pipeline see.
            self.scene.set_option("pipeline", 0)
            for _ in range(10):
                self.step()
            self.fence()
            k2 = self.scene.kernel_ms_history(8)
            kernel_alone_ms = float(k2.mean()) if len(k2) else None
            if launch is None and kernel_alone_ms is not None:
                kernel_ms = kernel_alone_ms
            self.scene.set_option("pipeline", 1)
            self.scene.stats()
        if env["world"] > 1:
            t = torch.tensor([dt, rays, kernel_ms, rays_ref], dtype=torch.float64, device="cpu" if env["rehearse"] else "cuda")
            tmax, tsum = t.clone(), t.clone()
            self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
            self.dist.all_reduce(tsum, op=self.dist.ReduceOp.SUM)
            dt, rays, kernel_ms, rays_ref = float(tmax[0]), float(tsum[1]), float(tmax[2]), float(tsum[3])
        return dict(dt=dt, rays=rays, rays_ref=rays_ref, kernel_ms=kernel_ms, steps=steps, tuning_frames=self.tuning_frames,
                    ramp_frames=self.ramp_frames, ramp_ms=self.ramp_ms, launch=launch, kernel_alone_ms=kernel_alone_ms,
                    overlapped=overlapped, refilled=refilled, sample_sync=int(self.scene.get_option("sample_sync_eff")),
                    pmode=self.scene.get_option("pmode"), merged_eff=self.scene.get_option("merged_eff"),
                    render_mode=self.scene.get_option("render_mode"))

    def close(self):
        self.scene.close()
        if self.presenter is not None:
            self.presenter.close()
        self.tiles = self.views = None


def kernel_name(m):
    """The kernel the timed frames ran, from what the library reports about its last launch (ptrt_get_option)."""
    if m["render_mode"] == 1:
        return "wf_trace_kernel + wf_shade_kernel (wavefront stages)"
    if m["render_mode"] == 2:
        return "path_trace_async_kernel (asynchronous lanes)"
    shape = {0: "lock-step", 1: "pairs over LDS-staged triangles", 2: "pair queue, separate shadow phase",
             3: "TLAS rounds", 4: "pair queue, shadow rays merged into the next traversal"}[m["pmode"]]
    if m.get("refilled"):
        return (f"path_trace_kernel<.., STREAM = true> PMODE {m['pmode']} ({shape}; megakernel of persistent waves with lane refill) + "
                "tonemap_tiles_kernel")
    return f"path_trace_kernel PMODE {m['pmode']} ({shape}; megakernel, fused tonemap)"


def roofline_block(config_name, m, pixels, ms_per_step):
    """HBM entry per the bench contract -- algorithmic bytes of ONE launch of the dominant kernel / that launch's duration,
    measured by HIP events on the stream it ran on, over the timed frames / peak -- plus the bound that binds this path: VALU
    issue.  Instruction counts, lane occupancy and measured HBM traffic come from the committed profile summary of the kernel
    the timed frames ran (profiles/summarize.py -> profiles/rNN_roofline_inputs.json: block `<config>_refill` for the
    lane-refill kernel, `<config>` for the one-tile-per-wave kernel)."""
    launch, kernel_ms = m.get("launch"), m["kernel_ms"]
    per_frame = launch["per_frame"] if launch else 1  # launches a frame was dealt to (they run concurrently)
    algo_bytes = ALGO_BYTES_PER_PIXEL * pixels / per_frame
    ok = kernel_ms == kernel_ms and kernel_ms > 0
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if ok else None
    prof, key = {}, None
    try:  # the newest round's summary (profiles/summarize.py)
        import glob
        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_roofline_inputs.json")))[-1]
        allp = json.load(open(newest))
        key = (config_name + "_refill") if (config_name and m.get("refilled")) else config_name
        prof = allp.get(key, {}) if key else {}
    except Exception:
        prof = {}
    valu = prof.get("valu_wave_instructions_per_launch")  # (the profile's launch covers the whole frame: --no-pipeline)
    traffic = prof.get("hbm_bytes_per_launch")
    # VALU issue: wave-instructions of a FRAME x 2 cycles over the time the chip spent on it -- the frame interval when frames
    # overlap (its launches share the chip with the neighbouring frames'), the launch's duration otherwise
    basis_ms = ms_per_step if launch else kernel_ms
    block = {"bound": "hbm", "achieved": None if achieved is None else round(achieved, 3), "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 6),
             "traffic": None if traffic is None else round(traffic / per_frame),
             "kernel_ms": round(kernel_ms, 4) if ok else None,
             "algorithmic_bytes_per_launch": algo_bytes,
             "launches_per_frame": per_frame,
             "binding": "VALU issue + memory latency under divergence (SURVEY 8(d)); the HBM entry is the contract's, not the limit",
             "kernel_ms_note": ("mean duration of the timed frames' own trace-kernel launches (HIP events on the auxiliary stream each ran "
                                "on, ptrt_launch_ms_history): a frame is `launches_per_frame` launches over interleaved rows of tiles "
                                "that run concurrently with each other and with the neighbouring frames', so a launch lasts less than "
                                "the frame interval ms_per_step and `achieved` x launches_per_frame is the device's rate")
                               if launch else "duration of ONE launch over the whole frame, ordered behind the stream (HIP events)",
             "valu_issue_frac": round(valu * VALU_CYCLES / (basis_ms * 1e-3 * SIMDS * CLOCK_HZ), 4) if (valu and ok) else None,
             "valu_issue_basis": "frame interval (ms_per_step)" if launch else "kernel_ms",
             "valu_wave_instructions_per_frame": valu, "lane_busy": prof.get("lane_busy"),
             "profile": prof.get("source"), "profile_block": key if prof else None}
    if launch:
        block["launch"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in launch.items()}
        block["frame_achieved"] = round(ALGO_BYTES_PER_PIXEL * pixels / (ms_per_step * 1e-3) / 1e9, 3)  # bytes of a frame / frame interval
        block["frame_frac"] = round(block["frame_achieved"] / HBM_PEAK_GBS, 6)
    if m.get("kernel_alone_ms") is not None:
        block["kernel_alone_ms"] = round(m["kernel_alone_ms"], 4)  # one launch of the one-tile-per-wave kernel, a frame alone on the chip
    return block


def spawn_ranks(n):
    """`python3 bench.py --gpus N` without a launcher: runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <the same arguments>` as a child process, passes its output through
    (rank 0 prints the one JSON line) and returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:  # a free port for the rendezvous (several benches may share a host)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL between processes needs it on this image)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # (short runs are bimodal on a fresh box: the first ~30 frames after start-up can
    ap.add_argument("--warmup", type=int, default=40)   #  run at half speed -- fluid 1.0 vs 1.85 ms with 12 + 30 frames; 40 + 100 is 0.3 s)  # (the queue modes settle their loop shape on frames 4-8: ptrt_set_option merged)
    ap.add_argument("--config", default="cornell1080", choices=sorted(CONFIGS))
    ap.add_argument("--scene", default=None, help="override the config's scene (also: many, matrix)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--preset", default=None, choices=["fast", "performance", "balanced", "quality", "ultra"],
                    help="Scene::setPerformancePreset on the measured scene (N=1 only; not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs3", action="store_true", help="skip the extra configs[3] measurement of a default run")
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
    ap.add_argument("--layout", default="strips", choices=["strips", "bands"],
                    help="N > 1: interleaved 8-row strips per rank (default: balanced) or contiguous bands")
    ap.add_argument("--farm", type=int, default=0, metavar="PARTS",
                    help="N=1 only, not the headline: also time the single-process C++ TileFarm (ptrt_farm_*) with PARTS "
                         "parts over the visible devices and report config.farm_ms_per_frame")
    ap.add_argument("--fluid-source", default="device", choices=["device", "host", "commit"],
                    help="fluid scene: new vertex positions from HBM (default), from pinned host memory (H2D inside the step), or "
                         "through the reference's caller: updatePTScene's vertex rewrite + commitObjectChanges()")
    ap.add_argument("--via-commit", action="store_true", help="= --fluid-source commit")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="order every frame's launch behind the stream (ptrt_set_option pipeline=0): what the profile passes use, "
                         "so that a launch's duration is a frame's")
    ap.add_argument("--no-ramp", action="store_true", help="skip the clock-ramp guard (untimed frames until two 10-frame means agree within 2 %%)")
    ap.add_argument("--no-time-launches", action="store_true", help="no events around the timed frames' launches (A/B of their cost)")
    ap.add_argument("--farm-serial", action="store_true", help="--farm: enqueue the parts in a row from the calling thread (A/B)")
    ap.add_argument("--rebuild", action="store_true",
                    help="fluid scene: rebuild the water BVH on the GPU every frame (ptrt_build_bvh) instead of refitting it")
    args = ap.parse_args()
    if args.via_commit:
        args.fluid_source = "commit"
    cfg = dict(CONFIGS[args.config])
    for k in ("scene", "width", "height", "spp", "depth"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    plain = not (args.denoise or args.bloom or args.preset or args.opt or args.scale != 1.0 or args.rebuild or
                 args.fluid_source != "device")
    headline = cfg == CONFIGS["cornell1080"] and plain
    # the committed profile (instruction counts, lane occupancy, HBM traffic) that belongs to this very workload, if any
    profile_key = None
    if plain:
        for name, c in list(CONFIGS.items()) + [("many", dict(CONFIGS["cornell1080"], scene="many"))]:
            if c == cfg:
                profile_key = name

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started plainly (`python3 bench.py --gpus N`), not by torch.distributed.run: this process -- which has imported nothing
        # that touches a device -- starts the N ranks as a CHILD (never exec: the box refuses an exec from a process that has
        # initialised the GPU, and a child keeps this one free to relay), relays rank 0's line and exits with the child's code.
        sys.exit(spawn_ranks(args.gpus))

    if args.gpus > 1:
        # One rank holds the context's stream, its two auxiliary streams, torch's and RCCL's own.  The runtime maps streams onto
        # GPU_MAX_HW_QUEUES (4) in-order hardware queues; two launches of a frame that end up on one queue run one after the other
        # (seen once on one GPU with a fourth busy stream).  Eight queues: room for all.  Set before HIP starts.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import ptrt_amd as P

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: torch.distributed.run was started with another --nproc-per-node")
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
    env = dict(rank=rank, world=world, dev=dev_index, rehearse=rehearse)

    W, H = cfg["width"], cfg["height"]
    farm = Farm(P, torch, dist, tilefarm, env, cfg["scene"], W, H, cfg["spp"], cfg["depth"], args=args, preset=args.preset)
    m = farm.measure(args.steps, args.warmup, ramp=not args.no_ramp, time_launches=not args.no_time_launches)
    settings = farm.scene.settings()
    fluid_sources = None
    if cfg["scene"] == "fluid" and world == 1 and not args.rebuild:
        # configs[4] three ways (the line's value is --fluid-source, default "device"): ms per frame of the same pipeline with
        # the positions already in HBM, crossing PCIe from pinned memory, and through the reference's caller + commit
        fluid_sources = {args.fluid_source + "_ms": round(m["dt"] / args.steps * 1e3, 4)}
        for src in ("device", "host", "commit"):
            if src != args.fluid_source:
                farm.fluid_source = src
                ms = farm.measure(max(4, args.steps // 2), 3, ramp=False)
                fluid_sources[src + "_ms"] = round(ms["dt"] / ms["steps"] * 1e3, 4)
        farm.fluid_source = args.fluid_source
        fluid_sources["gpu_commits, geometry_uploads"] = list(farm.scene.commitCounts())
        # host time of one commit-path frame, split: the reference caller's rewrite (setTriangleSoup = what updatePTScene does
        # to a `Triangles` mesh: 393,216 push_backs + the local box) and the mirror's commitObjectChanges() (topology check,
        # vertex hand-over through pinned staging, refit launches); measured over a few more frames, GPU idle-waited per frame
        farm.fluid_source = "commit"
        c0 = farm.scene.commitHostMicros()
        n_c = 12
        for _ in range(n_c):
            farm.step()
        farm.fence()
        c1 = farm.scene.commitHostMicros()
        farm.fluid_source = args.fluid_source
        fluid_sources["commit_host_us"] = {"caller": round((c1[0] - c0[0]) / n_c, 1), "mirror": round((c1[1] - c0[1]) / n_c, 1),
                                           "mirror_face_compare": round((c1[2] - c0[2]) / n_c, 1)}
        fluid_sources["note"] = ("device: positions resident in HBM; host: + 4.7 MB H2D from pinned memory per step; commit: the "
                                 "reference's caller (updatePTScene's rewrite of mesh->vertices/faces on the host, dirty flags, "
                                 "commitObjectChanges()) under Scene::setDynamicGeometryPolicy(GpuRefit) -- its host work "
                                 "(393,216 push_backs, the local box, pageable H2D) is the reference's, not the back end's")
    present = None
    if args.present > 0 and world == 1:
        farm.scene.set_stream(0)  # the viewer loop runs on the context's own stream
        farm.scene.view_run(args.warmup + 2, slots=args.present, keep=False)
        _, present = farm.scene.view_run(args.steps, slots=args.present, keep=False)
    rows0 = farm.rows
    post_on_rank0 = farm.post_on_rank0
    layout = "strips" if farm.strips else "bands"
    farm.close()
    farm_ms = farm_transport = None
    if args.farm > 0 and world == 1:  # the C++ farm below the C ABI: one process, its parts cycled over the visible devices
        ndev = torch.cuda.device_count()
        tf_ = P.TileFarm(W, H, [i % ndev for i in range(args.farm)], strips=(args.layout == "strips"))
        for sc in tf_.scenes:
            getattr(P.scenes, {"matrix": "material_matrix"}.get(cfg["scene"], cfg["scene"]))(sc)
            sc.setPerfSamplesPerPixel(cfg["spp"])
            sc.setMaxBounceDepth(cfg["depth"])
            sc.setDenoiserEnabled(False)
            sc.setBloomEnabled(False)
            sc.initBlueNoise()
            sc.uploadToGPU()
        if args.farm_serial:
            tf_.set_parallel(False)
        target = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
        for _ in range(args.warmup):
            tf_.render_to_device(target.data_ptr())
        tf_.sync()
        t0 = time.perf_counter()
        farm_host = []
        for _ in range(args.steps):
            tf_.render_to_device(target.data_ptr())
            farm_host.append(tf_.host_us)  # the calling thread's time inside the frame (parts enqueued on worker threads)
        tf_.sync()
        for d in range(ndev):
            torch.cuda.synchronize(d)
        farm_ms = (time.perf_counter() - t0) / args.steps * 1e3
        farm_transport = tf_.transport
        tf_.close()

    # configs[3] (showcase 3840x2160, 8 spp) over the same ranks, a few frames: the curve BASELINE asks for
    c3 = None
    if headline and not args.no_configs3:
        c = CONFIGS["showcase4k8"]
        n3 = max(2, min(args.steps, 6))
        f3 = Farm(P, torch, dist, tilefarm, env, c["scene"], c["width"], c["height"], c["spp"], c["depth"])
        m3 = f3.measure(n3, 9)  # (+ the frames on which the showcase kernel picks its loop shape: Farm.measure)
        part3 = "interleaved 8-row strip set(s)" if f3.strips else "band(s)"
        f3.close()
        c3 = {"workload": f"showcase {c['width']}x{c['height']} {c['spp']}spp {c['depth']}-bounce, {world} {part3}",
              "metric": "Mrays/s", "value": round(m3["rays"] / m3["dt"] / 1e6, 2), "n_gpus": world, "steps": n3, "warmup": 9,
              "ms_per_step": round(m3["dt"] / n3 * 1e3, 4), "fps": round(n3 / m3["dt"], 3), "scaling": "strong",
              "rays_per_frame": round(m3["rays"] / n3), "kernel_ms_max_over_ranks": round(m3["kernel_ms"], 4),
              "tuning_frames": m3["tuning_frames"], "ramp_frames": m3["ramp_frames"], "kernel": kernel_name(m3)}

    if rank != 0:
        dist.destroy_process_group()
        return

    dt, rays, kernel_ms = m["dt"], m["rays"], m["kernel_ms"]
    ms_per_step = dt / args.steps * 1e3
    spp_used, depth_used = settings["spp"], settings["depth"]
    out = {
        "metric": "Mrays/s", "value": round(rays / dt / 1e6, 2), "unit": "Mrays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        **({"rehearsal": "gloo + host staging on one GPU: NOT a measurement"} if rehearse else {}),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "fps": round(args.steps / dt, 2),
        "rays_per_frame": round(rays / args.steps),
        "rays_reference_path": round(m["rays_ref"] / args.steps),
        "rays_note": "value counts rays actually traced: extension rays + WALKED shadow rays (ptrt_stats.shadow_rays_walked; the "
                     "oracle counts the same).  rays_reference_path also counts the light samples whose value is exactly zero: "
                     "the reference sends a shadow ray for them, this kernel does not (DESIGN.md 3.1)",
        "config": {"workload": f"{cfg['scene']} {W}x{H} {spp_used}spp {depth_used}-bounce"
                               + (f" preset {args.preset}" if args.preset else "")
                               + (" +denoise" if args.denoise else "") + (" +bloom" if args.bloom else "")
                               + (f" scale{args.scale}" if args.scale != 1.0 else "")
                               + (" +gpu-rebuild" if args.rebuild else ""), "name": args.config, "scene": cfg["scene"],
                   "width": W, "height": H, "spp": spp_used, "max_depth": depth_used,
                   "parallelism": (f"tile{world}-{layout}" + ("+post-on-rank0" if post_on_rank0 else "")) if world > 1 else "single",
                   "kernel": kernel_name(m), "tuning_frames": m["tuning_frames"], "ramp_frames": m["ramp_frames"], "ramp_block_ms": m["ramp_ms"],
                   "library": P.library_info(),
                   "frames_overlap": m["overlapped"], "samples_in_step": bool(m.get("sample_sync")),
                   "pmode": int(m["pmode"]), "merged_eff": int(m["merged_eff"]), "lane_refill": bool(m["refilled"])},
        "roofline": roofline_block(profile_key if world == 1 else None, m, W * rows0, ms_per_step),
    }
    if c3 is not None:
        out["configs3"] = c3
    if fluid_sources is not None:
        out["config"]["fluid_sources"] = fluid_sources
        out["config"]["fluid_source"] = args.fluid_source
    if farm_ms is not None:
        out["config"]["farm_ms_per_frame"] = round(farm_ms, 4)
        out["config"]["farm"] = (f"{args.farm} parts, {args.layout}, one process, transport {farm_transport}, parts enqueued "
                                 + ("in a row" if args.farm_serial else "from one worker thread each"))
        out["config"]["farm_host_us_per_frame"] = round(sorted(farm_host)[len(farm_host) // 2], 1)
    if present is not None:
        out["config"]["present_ms_per_frame"] = round(present, 4)
        out["config"]["present_slots"] = args.present
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(P, cfg["scene"], W, H, spp_used, depth_used, 0,
                                           min(16, len(os.sched_getaffinity(0))))
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

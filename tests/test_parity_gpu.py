"""GPU parity proper: the HIP path (through the C ABI) against the CPU oracle, bit for bit.
PARITY UNPINNED w.r.t. the CUDA reference itself (see oracle/ptrt_oracle.cpp)."""
import numpy as np
import pytest

from common import assert_frames_equal, render_both

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size,spp,depth,frames", [((64, 64), 1, 1, 1), ((64, 64), 1, 2, 1), ((64, 64), 1, 4, 2),
                                                   ((256, 256), 1, 1, 1), ((96, 72), 4, 4, 3), ((61, 45), 2, 3, 1)])
def test_cornell_bit_exact(P, O, blue_noise, size, spp, depth, frames):
    s = P.Scene(size[0], size[1])
    P.scenes.cornell(s)
    gpu, cpu = render_both(P, O, s, blue_noise, spp, depth, frames)
    assert_frames_equal(gpu, cpu)
    if depth == 1:  # no first-hit NEE, emitter out of view -> black, 1 ray per pixel
        assert not gpu[0]["accum"].any()
        assert gpu[0]["stats"]["extension_rays"] == size[0] * size[1] * spp
        assert gpu[0]["stats"]["shadow_rays"] == 0
    s.close()


@pytest.mark.parametrize("size,spp,depth,frames,kw", [
    ((96, 72), 4, 4, 3, {}), ((61, 45), 2, 3, 2, {}), ((1, 1), 3, 4, 2, {}), ((250, 141), 2, 5, 2, {}),
    ((64, 64), 2, 4, 2, dict(tile_y0=24, tile_rows=16)),          # a band context: the queue covers its rows only
    ((96, 72), 2, 4, 2, dict(force_full=1)),                       # the all-materials variant
    ((640, 360), 4, 4, 1, {})])                                    # more tiles than persistent waves would draw at once
def test_cornell_lane_refill(P, O, blue_noise, size, spp, depth, frames, kw):
    """ptrt_set_option "refill", 2: PMODE 1 as persistent waves whose lanes take the launch's next pixel when theirs is
    finished (path_trace_kernel<.., STREAM = true> + tonemap_tiles_kernel) -- every buffer, generator state and ray count
    as the oracle has them.  (By default only frames that overlap their predecessor run this way: test_misc_gpu.py.)"""
    kw = dict(kw)
    force_full = kw.pop("force_full", 0)
    s = P.Scene(size[0], size[1], **kw)
    P.scenes.cornell(s)
    s.set_option("refill", 2)
    s.set_option("persist", 1 if size == (640, 360) else 0)  # (256 waves for 3,600 tiles: every wave draws many tickets)
    if force_full:
        s.set_option("force_full", 1)
    gpu, cpu = render_both(P, O, s, blue_noise, spp, depth, frames)
    assert s.get_option("refilled") == 1
    assert_frames_equal(gpu, cpu)
    s.close()


@pytest.mark.parametrize("sync", [0, 1])
@pytest.mark.parametrize("scene,size,spp,depth", [("cornell", (96, 72), 4, 4), ("cornell", (61, 45), 3, 7), ("showcase", (80, 64), 2, 4),
                                                  ("many", (72, 48), 2, 5)])
def test_samples_in_step_or_not(P, O, blue_noise, sync, scene, size, spp, depth):
    """ptrt_set_option "sample_sync" forced both ways (the default follows the depth limit and the traversal mode): the lanes
    of a wave start their samples together, or each as soon as its path has ended -- the oracle's frames either way."""
    s = P.Scene(size[0], size[1])
    if scene == "many":
        _many_meshes(P, s, n=40)
    else:
        (P.scenes.cornell if scene == "cornell" else (lambda sc: P.scenes.showcase(sc, segments=8)))(s)
    s.set_option("sample_sync", sync)
    gpu, cpu = render_both(P, O, s, blue_noise, spp, depth, 2)
    assert s.get_option("sample_sync_eff") == sync
    assert_frames_equal(gpu, cpu)
    s.close()


def test_lane_refill_queue_shapes(P):
    """The tile queue of the lane-refill kernel at its edges: frames of a few pixels to a few hundred tiles with ragged right
    and bottom edges, one persistent wave to more waves than tiles, tickets good for 1..5 tiles (a last ticket that reaches past
    the launch), 1 spp (a lane finishes a pixel every iteration) and long paths.  Every buffer, generator state and ray count
    equals the one-tile-per-wave kernel's (which the other tests hold to the oracle), over three frames."""
    import random
    rnd = random.Random(20261005)
    cases = [(1, 1), (8, 8), (9, 1), (1, 9), (64, 8), (65, 9)] + [(rnd.randint(2, 150), rnd.randint(2, 90)) for _ in range(18)]
    for k, (W, H) in enumerate(cases):
        spp, depth = rnd.choice([(1, 1), (1, 4), (2, 3), (3, 5), (4, 4)])
        opts = dict(persist=rnd.choice([0, 1, 1, 2, 3]), ticket_tiles=rnd.choice([1, 1, 2, 3, 5]))
        frames = {}
        for refill in (0, 2):
            s = P.Scene(W, H)
            (P.scenes.cornell if k % 3 else (lambda sc: P.scenes.cornell(sc, quads=True)))(s)
            s.setPerfSamplesPerPixel(spp)
            s.setMaxBounceDepth(depth)
            s.setDenoiserEnabled(False)
            s.setBloomEnabled(False)
            s.initBlueNoise()
            s.uploadToGPU()
            s.set_option("count_rays", 1)
            s.set_option("refill", refill)
            for name, v in opts.items():
                s.set_option(name, v)
            out = []
            for _ in range(3):
                rgb = s.render_to_host()
                assert s.get_option("refilled") == (1 if refill else 0)
                out.append((rgb, s.read(P.BUF_ACCUM), s.read(P.BUF_RNG), s.read(P.BUF_NORMAL), s.read(P.BUF_DEPTH),
                            s.read(P.BUF_OBJECT_ID), s.stats()))
            frames[refill] = out
            s.close()
        for f, (a, b) in enumerate(zip(frames[0], frames[2])):
            for i in range(6):
                av, bv = (a[i].view(np.uint32), b[i].view(np.uint32)) if a[i].dtype == np.float32 else (a[i], b[i])
                assert np.array_equal(av, bv), f"case {k} {W}x{H} spp {spp} depth {depth} {opts} frame {f}: buffer {i} differs"
            assert a[6] == b[6], f"case {k} {W}x{H} {opts} frame {f}: {a[6]} vs {b[6]}"


def test_lane_refill_over_interleaved_strips_and_into_a_shared_frame(P):
    """Lane refill on contexts that own every third 8-row strip, each writing its rows straight into ONE device frame
    (PTRT_OUT_DEVICE_FRAME): the frame and the strips' HDR rows are the full-frame context's (classic kernel)."""
    import torch
    W, H = 88, 52
    def prep(s, refill):
        P.scenes.cornell(s)
        s.setPerfSamplesPerPixel(3)
        s.setMaxBounceDepth(4)
        s.setDenoiserEnabled(False)
        s.setBloomEnabled(False)
        s.initBlueNoise()
        s.uploadToGPU()
        s.set_option("refill", refill)
    full = P.Scene(W, H)
    prep(full, 0)
    want = [full.render_to_host() for _ in range(2)]
    want_acc = full.read(P.BUF_ACCUM).reshape(H, W, 3)
    full.close()
    frame = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    farm = P.TileFarm(W, H, [0] * 3, strips=True)  # (the parts on the presenting GPU write straight into the frame)
    for s in farm.scenes:
        prep(s, 2)
    for f in range(2):
        farm.render_to_device(frame.data_ptr())
        torch.cuda.synchronize()
        assert all(s.get_option("refilled") == 1 for s in farm.scenes)
        assert np.array_equal(frame.cpu().numpy(), want[f])
    for r, s in enumerate(farm.scenes):
        rows = [y for y in range(H) if (y // 8) % 3 == r]
        assert np.array_equal(s.read(P.BUF_ACCUM).reshape(-1, W, 3).view(np.uint32), want_acc[rows].view(np.uint32))
    farm.close()


@pytest.mark.parametrize("force_geom", [1, 2])
def test_cornell_general_traversal_variants(P, O, blue_noise, force_geom):
    """The general BLAS / general TLAS kernels must give the brute-force variant's bits."""
    s = P.Scene(80, 64)
    P.scenes.cornell(s)
    s.set_option("force_geom", force_geom)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 2)
    assert_frames_equal(gpu, cpu)
    s.close()


def test_cornell_lockstep_variant(P, O, blue_noise):
    """pair_trace=0: the lock-step mesh loop (no (ray, mesh) pair compaction) gives the same bits."""
    s = P.Scene(88, 72)
    P.scenes.cornell(s)
    s.set_option("pair_trace", 0)
    gpu, cpu = render_both(P, O, s, blue_noise, 3, 4, 2)
    assert_frames_equal(gpu, cpu)
    s.close()


def test_instanced_meshes(P, O, blue_noise):
    """has_transform instances (setPosition/setRotation/scale on the instance transform), incl. the
    reference's inverse-matrix quirk, in the single-leaf (pair) and deep-BVH kernels."""
    for leaf in ((12, 5), (2, 0)):
        s = P.Scene(72, 64)
        P.scenes.cornell(s)
        extra = s.addCube(P.Material((0.2, 0.3, 0.9), 0.4))
        s.setPosition(extra, (1.0, -1.0, -5.0))
        s.setRotation(extra, (0.3, 0.5, 0.1))
        s.setInstanceScale(extra, (1.5, 0.7, 1.2))
        ball = s.addSphere(6, P.Material((0.9, 0.9, 0.2), 0.05, 1.0))
        s.setPosition(ball, (-2.0, 1.5, -4.0))
        s.setBVHLeafTarget(*leaf)
        gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 2)
        assert_frames_equal(gpu, cpu)
        assert (gpu[0]["object_id"] == extra).sum() > 50
        s.close()


@pytest.mark.parametrize("leaf,pair_trace", [((1, 0), 1), ((2, 1), 1), ((4, 0), 1), ((2, 1), 0)])
def test_cornell_deep_bvh(P, O, blue_noise, leaf, pair_trace):
    """Small leaf targets turn every cube into a multi-level BLAS and the scene into a real TLAS."""
    s = P.Scene(72, 56)
    P.scenes.cornell(s)
    s.setBVHLeafTarget(*leaf)
    s.set_option("pair_trace", pair_trace)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 2)
    assert_frames_equal(gpu, cpu)
    s.close()


def test_cornell_one_lane_per_pair(P, O, blue_noise):
    """pair_split=0: partial pair batches keep one lane per pair (the default gives a pair 2^k lanes there)."""
    s = P.Scene(88, 72)
    P.scenes.cornell(s)
    s.set_option("pair_split", 0)
    gpu, cpu = render_both(P, O, s, blue_noise, 3, 4, 2)
    assert_frames_equal(gpu, cpu)
    s.close()


@pytest.mark.parametrize("stage,lights,pad", [(0, 1, 0), (4, 1, 0), (5, 3, 0), (6, 8, 0), (7, 8, 0), (7, 11, 0), (7, 3, 4096)])
def test_cornell_staged_shading_inputs(P, O, blue_noise, stage, lights, pad):
    """PMODE 1 keeps its shading inputs in LDS (jitter table, blue-noise values, light and material records): every subset,
    more lights than fit (11 > 8: they stay in global memory), and a launch whose LDS has no room (nothing staged)."""
    s = P.Scene(88, 72)
    P.scenes.cornell(s)
    for i in range(1, lights):  # (the scene has one light already)
        if i % 3 == 0:
            s.addSpotLight((0.3 * i - 1.0, 1.6, 0.2 * i - 1.0), (0.1, -1.0, 0.05 * i), (1.0, 0.8, 0.6), 6.0, 0.6, 0.9, 50.0, 0.05)
        elif i % 3 == 1:
            s.addPointLight((0.2 * i - 1.0, 1.2, -0.5 + 0.1 * i), (0.5, 0.7, 1.0), 3.0, 30.0, 0.1 if i % 2 else 0.0)
        else:
            s.addDirectionalLight((0.3, -1.0, 0.2 * i), (0.9, 0.9, 0.8), 0.7)
    s.set_option("stage", stage)
    s.set_option("lds_pad", pad)
    gpu, cpu = render_both(P, O, s, blue_noise, 3, 4, 2)
    assert_frames_equal(gpu, cpu)
    s.close()


def _many_meshes(P, s, n=40, instanced=True):
    P.scenes.many(s, n, instanced)


@pytest.mark.parametrize("pair_trace,leaf", [(1, None), (0, None), (1, (2, 0)), (1, (4, 1))])
def test_real_tlas_many_meshes(P, O, blue_noise, pair_trace, leaf):
    """48 meshes -> a TLAS with inner nodes: PMODE 3 (per-lane TLAS walk, one leaf of (ray, mesh) pairs per round)
    and, with pair_trace=0, the lock-step general traversal; also with small BLAS/TLAS leaf targets."""
    s = P.Scene(96, 64)
    _many_meshes(P, s)
    if leaf:
        s.setBVHLeafTarget(*leaf)
    s.set_option("pair_trace", pair_trace)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 2)
    assert_frames_equal(gpu, cpu)
    assert len(set(gpu[0]["object_id"].tolist())) > 20
    s.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_real_tlas_of_hostile_instances(P, O, blue_noise, seed):
    """PMODE 3's conservative world-space pre-test of instance root boxes (ptrt_capi.hip upload_instance_pretests) must
    never reject a ray the reference's local-space test accepts.  56 instances only -- thin slabs, needles, large and tiny
    scales, every rotation, translations that hit the reference's wrong mat4::inverse (rotation + x translation), some far
    from the origin, boxes that touch and overlap -- and a camera INSIDE the cloud, so rays start in and next to the boxes;
    3 frames x 3 spp x 6 bounces against the oracle, every buffer and ray count."""
    rs = np.random.RandomState(seed)
    s = P.Scene(80, 56)
    s.addPlaneXZ(-3.0, 30.0, P.Material((0.5, 0.5, 0.5), 0.7))
    for k in range(56):
        mat = P.Material(tuple(rs.uniform(0.2, 0.9, 3)), float(rs.uniform(0.02, 0.9)), float(k % 3 == 0),
                         transmission=1.0 if k % 9 == 4 else 0.0, ior=1.3)
        m = s.addSphere(4, mat) if k % 4 == 1 else s.addCube(mat)
        shape = k % 4
        scale = {0: rs.uniform(0.3, 1.5, 3), 1: rs.uniform(0.05, 0.2, 3), 2: (rs.uniform(2.0, 5.0), 0.02, rs.uniform(0.5, 2.0)),
                 3: (0.03, rs.uniform(1.0, 4.0), 0.03)}[shape]
        far = 40.0 if k % 11 == 7 else 0.0
        s.setPosition(m, (float(rs.uniform(-4, 4) + far), float(rs.uniform(-2.5, 3)), float(rs.uniform(-9, 1))))
        s.setRotation(m, tuple(rs.uniform(-3.1, 3.1, 3)))
        s.setInstanceScale(m, tuple(float(v) for v in scale))
    s.addPointLight((0, 4, -3), (1.0, 0.95, 0.9), 6.0, 30.0, 0.3)
    s.addDirectionalLight((-0.3, -1.0, -0.2), (0.9, 0.9, 1.0), 1.5)
    s.setSkyGradient((0.5, 0.6, 0.9), (0.9, 0.9, 0.9))
    s.setCamera((0.3, 0.2, -3.5), (0.0, 0.0, -8.0), (0, 1, 0), 75.0)
    gpu, cpu = render_both(P, O, s, blue_noise, 3, 6, 3)
    assert_frames_equal(gpu, cpu)
    assert len(set(gpu[0]["object_id"].tolist())) > 5 and gpu[0]["stats"]["shadow_rays"] > 0
    s.close()


@pytest.mark.parametrize("size", [(1, 1), (7, 3), (8, 8), (9, 17)])
def test_tiny_and_ragged_frames(P, O, blue_noise, size):
    """Frames smaller than one 8x8 tile and frames whose edges cut tiles."""
    for build in (P.scenes.cornell, lambda s: P.scenes.showcase(s, segments=6)):
        s = P.Scene(*size)
        build(s)
        gpu, cpu = render_both(P, O, s, blue_noise, 2, 3, 2)
        assert_frames_equal(gpu, cpu)
        s.close()


@pytest.mark.parametrize("size", [(100, 60), (520, 24), (64, 520)])
@pytest.mark.parametrize("run", [0, 1, 3, 8, 64])
def test_workgroup_to_tile_maps_cover_the_frame(P, O, blue_noise, size, run):
    """Option tile_run (path_trace_kernel: of every 8 * run consecutive tiles workgroup p renders the tile XCD p % 8's run holds):
    a permutation of the launch's tiles whatever the frame's size -- launches of fewer tiles than one span, spans that cross
    rows of tiles, a trailing part that keeps its numbers --, so every pixel is rendered exactly once: the oracle's frame."""
    for build, opts in ((P.scenes.cornell, {}), (lambda s: P.scenes.showcase(s, segments=6), dict(merged=0)),
                        (lambda s: P.scenes.showcase(s, segments=6), dict(merged=1))):
        s = P.Scene(*size)
        build(s)
        s.set_option("tile_run", run)
        for k, v in opts.items():
            s.set_option(k, v)
        gpu, cpu = render_both(P, O, s, blue_noise, 1, 3, 1)
        assert_frames_equal(gpu, cpu)
        s.close()


def test_maximum_samples_and_bounces(P, O, blue_noise):
    """samplesPerPixel and maxBounceDepth at the setters' upper clamp (16, scene.cuh:1894-1897), a scene without
    lights (no light sampling at all) and one without sky."""
    s = P.Scene(24, 16)
    P.scenes.showcase(s, segments=8)
    gpu, cpu = render_both(P, O, s, blue_noise, 16, 16, 1)
    assert_frames_equal(gpu, cpu)
    assert gpu[0]["stats"]["paths"] == 24 * 16 * 16
    s.close()
    s = P.Scene(40, 32)
    for k in range(3):
        m = s.addCube(P.Material((0.8, 0.3 + 0.2 * k, 0.2), 0.4, emission=(2.0, 2.0, 2.0) if k == 1 else (0.0, 0.0, 0.0)))
        s.scale(m, (1.0 + k, 1.0, 1.0))
        s.moveTo(m, (-3.0 + 3.0 * k, -1.0 + k, -6.0))
    s.disableSky()
    gpu, cpu = render_both(P, O, s, blue_noise, 3, 5, 2)   # no lights were added
    assert_frames_equal(gpu, cpu)
    assert gpu[0]["stats"]["shadow_rays"] == 0 and gpu[0]["accum"].any()
    s.close()


def test_large_leaf_target(P, O, blue_noise):
    """A leaf target beyond the reference's default (12 + 5): leaves of up to 40 triangles do not fit the compacted
    leaf phase's test list, so the kernel walks them lane by lane -- same bits."""
    s = P.Scene(80, 56)
    P.scenes.showcase(s, segments=10)
    s.setBVHLeafTarget(30, 10)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 1)
    assert_frames_equal(gpu, cpu)
    s.set_option("async_lanes", 1)          # refused for such leaves: falls back to the default kernel
    s.render_to_host()
    import ctypes
    P.lib.ptrt_debug_last_render_mode.argtypes = [ctypes.c_void_p]
    assert P.lib.ptrt_debug_last_render_mode(s.ctx) == 0
    s.close()


def test_cornell_quads(P, O, blue_noise):
    s = P.Scene(64, 64)
    P.scenes.cornell(s, quads=True)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 1)
    assert_frames_equal(gpu, cpu)
    s.close()


def test_showcase_small(P, O, blue_noise):
    """All material branches (transmission, clearcoat, iridescence, sheen), spot + sphere lights, sky."""
    s = P.Scene(96, 64)
    P.scenes.showcase(s, segments=12)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 5, 2)
    assert_frames_equal(gpu, cpu)
    s.close()


def test_loop_shape_chosen_by_measurement(P, O, blue_noise):
    """merged = -1 (the default): the queue mode's two loop shapes take turns on a scene's frames 4-7 and the library keeps
    the faster from frame 8 on -- eleven consecutive frames, whichever shape each one ran in, equal the oracle's."""
    s = P.Scene(96, 64)
    P.scenes.showcase(s, segments=12)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 5, 11)
    assert_frames_equal(gpu, cpu)
    s.close()
    s = P.Scene(80, 56)  # (another setting of the queue: its own choice)
    P.scenes.showcase(s, segments=10)
    s.set_option("steal", 2)
    s.set_option("fetch_min", 32)
    gpu, cpu = render_both(P, O, s, blue_noise, 1, 4, 9)
    assert_frames_equal(gpu, cpu)
    s.close()


@pytest.mark.parametrize("merged", [1, 0, 2])
@pytest.mark.parametrize("fetch_min,leaf_pairs,leaf_min,steal", [(0, 0, 64, 0), (1, 1, 1, 1), (16, 0, 8, 2), (16, 1, 8, 0),
                                                                 (16, 1, 4, 2), (48, 1, 64, 8), (64, 1, 24, 1)])
def test_pair_queue_refill_thresholds(P, O, blue_noise, fetch_min, leaf_pairs, leaf_min, steal, merged):
    """PMODE 4 / PMODE 2 (deep BLASes behind a single-leaf TLAS; merged=1: one traversal per iteration, the parked
    shadow rays riding with the next extension rays -- needs leaf_pairs; merged=0: closest-hit and any-hit phases):
    static 64-pair batches (0) and the dynamic
    refill at every threshold, with the leaf phase lane by lane (0) or as compacted (lane, triangle)
    pairs (1), the node loop ending once leaf_min lanes wait at a leaf, and idle lanes stealing shadow-ray subtrees (steal > 0),
    give the oracle's bits -- showcase materials, plus instanced meshes."""
    if merged == 1 and not leaf_pairs:
        pytest.skip("the merged traversal always uses the compacted leaf phase")
    lds_nodes, merged = (1, 0) if merged == 2 else (0, merged)  # (2: the 4-wave workgroups with LDS-staged BLAS top levels)
    s = P.Scene(96, 64)
    P.scenes.showcase(s, segments=12)
    s.set_option("merged", merged)
    s.set_option("lds_nodes", lds_nodes)
    s.set_option("fetch_min", fetch_min)
    s.set_option("leaf_pairs", leaf_pairs)
    s.set_option("leaf_min", leaf_min)
    s.set_option("steal", steal)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 5, 2)
    assert_frames_equal(gpu, cpu)
    s.close()
    s = P.Scene(72, 64)
    P.scenes.cornell(s)
    extra = s.addCube(P.Material((0.2, 0.3, 0.9), 0.4))
    s.setPosition(extra, (1.0, -1.0, -5.0))
    s.setRotation(extra, (0.3, 0.5, 0.1))
    s.setInstanceScale(extra, (1.5, 0.7, 1.2))
    s.setBVHLeafTarget(2, 0)
    s.set_option("merged", merged)
    s.set_option("lds_nodes", lds_nodes)
    s.set_option("fetch_min", fetch_min)
    s.set_option("leaf_pairs", leaf_pairs)
    s.set_option("leaf_min", leaf_min)
    s.set_option("steal", steal)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 2)
    assert_frames_equal(gpu, cpu)
    s.close()


@pytest.mark.parametrize("csteal,csteal_min,leaf_pairs,leaf_min,fetch_min", [(1, 0, 1, 8, 16), (2, 0, 0, 64, 0), (4, 8, 1, 8, 16),
                                                                             (1, 2, 1, 1, 1), (8, 0, 1, 24, 64)])
@pytest.mark.parametrize("merged", [0, 1])
def test_closest_hit_stealing_is_exact(P, O, blue_noise, csteal, csteal_min, leaf_pairs, leaf_min, fetch_min, merged):
    """PMODE 2, option csteal: idle lanes take the bottom entry of a busy closest-hit walk's stack, with a copy of its ray and
    limit, and merge what they find; a ray for which a thief accepted a hit in front of its leaf box, or two walks of one pair
    reported the same distance, is traced again without stealing (run_closest_queue).  The oracle's bits at
    every threshold (csteal_min = 0: every walk is stolen from at once), with both leaf phases -- showcase materials, an
    instanced mesh with tiny leaves, and the water grid, whose triangle edges lie ON leaf-box faces (the case in which a
    looser limit finds a hit the sequential walk culls)."""
    if merged and not leaf_pairs:
        pytest.skip("the merged traversal always uses the compacted leaf phase")

    def opts(s):
        s.set_option("merged", merged)  # (1: PMODE 4, closest-hit and shadow pairs in one queue, both kinds stolen from)
        s.set_option("steal", 1)
        s.set_option("csteal", csteal)
        s.set_option("csteal_min", csteal_min)
        s.set_option("leaf_pairs", leaf_pairs)
        s.set_option("leaf_min", leaf_min)
        s.set_option("csteal_leaf_min", leaf_min)
        s.set_option("fetch_min", fetch_min)

    s = P.Scene(96, 64)
    P.scenes.showcase(s, segments=12)
    opts(s)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 5, 2)
    assert s.get_option("pmode") == (4 if merged else 2)
    assert_frames_equal(gpu, cpu)
    s.close()
    s = P.Scene(72, 64)
    P.scenes.cornell(s)
    extra = s.addCube(P.Material((0.2, 0.3, 0.9), 0.4))
    s.setPosition(extra, (1.0, -1.0, -5.0))
    s.setRotation(extra, (0.3, 0.5, 0.1))
    s.setInstanceScale(extra, (1.5, 0.7, 1.2))
    s.setBVHLeafTarget(2, 0)
    opts(s)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 2)
    assert_frames_equal(gpu, cpu)
    s.close()
    s = P.Scene(160, 96)
    P.scenes.fluid(s, cells=64, t=0.3)
    opts(s)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 2)
    assert s.get_option("pmode") == (4 if merged else 2)
    assert_frames_equal(gpu, cpu)
    s.close()


def test_tile_equals_full_frame(P, O, blue_noise):
    """Rows [24,40) rendered alone are the same bits as those rows of the full frame (global RNG keying)."""
    full = P.Scene(64, 64)
    P.scenes.cornell(full)
    g_full, _ = render_both(P, O, full, blue_noise, 2, 4, 1)
    tile = P.Scene(64, 64, tile_y0=24, tile_rows=16)
    P.scenes.cornell(tile)
    g_tile, c_tile = render_both(P, O, tile, blue_noise, 2, 4, 1)
    assert_frames_equal(g_tile, c_tile)
    sl = slice(24 * 64, 40 * 64)
    assert np.array_equal(g_tile[0]["accum"].view(np.uint32), g_full[0]["accum"][sl].view(np.uint32))
    assert np.array_equal(g_tile[0]["object_id"], g_full[0]["object_id"][sl])
    # RGB8 is bottom-up: tile row r (top-down) sits at full-image byte row H-1-(24+r)
    assert np.array_equal(g_tile[0]["rgb8"], g_full[0]["rgb8"][64 - 40:64 - 24])
    full.close()
    tile.close()


def test_trace_rays_known_answers(P, O):
    s = P.Scene(32, 32)
    P.scenes.cornell(s)
    s.uploadToGPU()
    rs = np.random.RandomState(7)
    n = 4096
    o = np.tile(np.array([[0, 0, -5]], np.float32), (n, 1)) + rs.uniform(-4, 4, (n, 3)).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    g = s.trace_rays(o, d)
    c = O.trace_rays(s.flatten(), o, d)
    for name in g.dtype.names:
        a, b = g[name], c[name]
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)
        assert np.array_equal(a, b), name
    assert g["hit"].sum() > n // 2
    s.close()

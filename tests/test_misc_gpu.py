"""GPU checks that are not whole-frame parity: deterministic math bit-equality, committed goldens,
error behaviour of the C ABI on a live device, size-independent properties at the benchmark size."""
import ctypes as C
import os

import numpy as np
import pytest

from common import assert_frames_equal, render_both

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_detmath_bits_match_oracle(P, O):
    s = P.Scene(16, 16)
    rs = np.random.RandomState(5)
    cases = [(0, rs.uniform(-50, 50, 1 << 16)), (1, rs.uniform(-50, 50, 1 << 16)), (0, np.linspace(0, 6.2831855, 1 << 16)),
             (2, rs.uniform(-110, 90, 1 << 16)), (3, np.exp(rs.uniform(-100, 88, 1 << 16))),
             (3, np.array([0.0, -1.0, np.inf, 1e-40, 1.0, np.nan])), (2, np.array([np.nan, -200.0, 100.0, 0.0]))]
    for op, x in cases:
        x = x.astype(np.float32)
        out = np.zeros_like(x)
        rc = P.lib.ptrt_debug_detmath(s.ctx, op, x.ctypes.data_as(C.POINTER(C.c_float)), None, x.size,
                                      out.ctypes.data_as(C.POINTER(C.c_float)))
        assert rc == 0
        assert np.array_equal(out.view(np.uint32), O.detmath(op, x).view(np.uint32)), op
    z = rs.uniform(0.003, 1.0, 1 << 16).astype(np.float32)
    y = np.full_like(z, np.float32(1.0 / 2.4))
    out = np.zeros_like(z)
    P.lib.ptrt_debug_detmath(s.ctx, 4, z.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                             z.size, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(out.view(np.uint32), O.detmath(4, z, y).view(np.uint32))
    s.close()


@pytest.mark.parametrize("name", ["cornell_64x64_1spp_d4_f0", "cornell_64x64_4spp_d2_f3", "showcase12_64x48_2spp_d5_f0"])
def test_gpu_reproduces_committed_goldens(P, name):
    import make_oracle_golden as M
    scene, w, h, spp, depth, frame = M.CASES[name]
    g = np.load(os.path.join(GOLD, f"oracle_{name}.npz"))
    s = P.Scene(w, h)
    P.scenes.cornell(s) if scene == "cornell" else P.scenes.showcase(s, segments=12)
    s.setPerfSamplesPerPixel(spp)
    s.setMaxBounceDepth(depth)
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    s.initBlueNoise()
    s.uploadToGPU()
    s.setFrameCount(frame)
    rgb = s.render_to_host()
    assert np.array_equal(s.read(P.BUF_OBJECT_ID), g["object_id"])
    assert np.array_equal(s.read(P.BUF_ACCUM).view(np.uint32), g["accum"].view(np.uint32))
    assert np.array_equal(s.read(P.BUF_DEPTH).view(np.uint32), g["depth"].view(np.uint32))
    assert np.array_equal(rgb, g["rgb8"])
    s.close()


def test_fast_reciprocal_is_ieee_for_every_float(P):
    """rcp_ieee (v_rcp_f32 + one Newton step, guarded by exponent) == 1.0f/y for all 2^32 inputs."""
    s = P.Scene(16, 16)
    out = (C.c_uint * 9)()
    P.lib.ptrt_debug_rcp_check.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    assert P.lib.ptrt_debug_rcp_check(s.ctx, out) == 0
    assert out[0] == 0, f"{out[0]} mismatches, first inputs: {[hex(v) for v in list(out)[1:9]]}"
    s.close()


def test_fast_square_root_is_ieee_for_every_float(P):
    """sqrt_ieee (v_rsq_f32 + one exact-residual correction, guarded by exponent) == sqrtf(x) for all 2^32 inputs."""
    s = P.Scene(16, 16)
    out = (C.c_uint * 9)()
    P.lib.ptrt_debug_sqrt_check.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    assert P.lib.ptrt_debug_sqrt_check(s.ctx, out) == 0
    assert out[0] == 0 and out[1] == 0, f"{out[0]} mismatches ({out[1]} of the bare core), first inputs: {[hex(v) for v in list(out)[2:9]]}"
    s.close()


def test_render_before_upload_and_bad_scenes(P):
    s = P.Scene(32, 32)
    assert P.lib.ptrt_render(s.ctx, 0, 1, 1, None, 0) == -4          # PTRT_E_NOT_READY
    assert b"geometry" in P.lib.ptrt_last_error(s.ctx)
    P.scenes.cornell(s)
    s.uploadToGPU()
    assert P.lib.ptrt_render(s.ctx, 0, 0, 4, None, 0) == -1          # spp < 1
    assert P.lib.ptrt_render(s.ctx, -1, 1, 4, None, 0) == -1         # a negative frame index would index the jitter table out of bounds
    assert P.lib.ptrt_render(s.ctx, 2**31 - 1, 4, 4, None, 0) == -1
    # a face index out of range is rejected at upload, not discovered by a faulting kernel
    d = s.flatten().contents
    bad = (P.Tri * 12)(*[P.Tri(0, 1, 99) for _ in range(12)])
    m0 = d.meshes[0]
    keep = m0.faces
    m0.faces = C.cast(bad, C.POINTER(P.Tri))
    rc = P.lib.ptrt_upload_geometry(s.ctx, d.meshes, d.mesh_count, d.tlas_nodes, d.tlas_node_count,
                                    d.tlas_mesh_indices, d.tlas_index_count)
    m0.faces = keep
    assert rc == -1 and b"vertex out of range" in P.lib.ptrt_last_error(s.ctx)
    s.close()


def test_full_size_properties(P):
    """1920x1080, 4 spp, 4 bounces (BASELINE configs[1]): properties that need no oracle run."""
    s = P.Scene(1920, 1080)
    P.scenes.cornell(s)
    s.setPerfSamplesPerPixel(4)
    s.setMaxBounceDepth(4)
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    s.initBlueNoise()
    s.uploadToGPU()
    s.set_option("count_rays", 1)
    rng0 = s.read(P.BUF_RNG)
    rgb = s.render_to_host()
    st = s.stats()
    acc, oid, dep, nrm, rng1 = (s.read(k) for k in (P.BUF_ACCUM, P.BUF_OBJECT_ID, P.BUF_DEPTH, P.BUF_NORMAL, P.BUF_RNG))
    n = 1920 * 1080
    assert st["paths"] == 4 * n and 4.5 < (st["extension_rays"] + st["shadow_rays"]) / st["paths"] < 5.3
    assert np.isfinite(acc).all() and acc.min() >= 0 and (0.2126 * acc[:, 0] + 0.7152 * acc[:, 1] + 0.0722 * acc[:, 2]).max() <= 100.0001
    assert (oid >= 0).all() and (oid < 8).all() and np.allclose(np.linalg.norm(nrm, axis=1), 1, atol=1e-5)
    draws = ((rng1[:, 0].astype(np.int64) - rng0[:, 0]) % (1 << 32)) // 362437
    assert draws.min() >= 4 * 3 and draws.max() <= 4 * 17           # Appendix C budget per sample
    assert np.array_equal(rng1[:, 0] - rng0[:, 0], (draws * 362437).astype(np.uint32))
    # left-right symmetry of the box geometry: object ids of walls mirror (1 <-> 2), others keep
    ids = oid.reshape(1080, 1920)
    mirror = ids[:, ::-1].copy()
    sw = mirror.copy()
    sw[mirror == 1], sw[mirror == 2] = 2, 1
    walls = np.isin(ids, (0, 3, 4)) & np.isin(sw, (0, 3, 4, 1, 2))
    assert (ids[walls] == sw[walls]).mean() > 0.99
    # the same frame rendered as two bands is the same bytes
    top = P.Scene(1920, 1080, tile_y0=0, tile_rows=536)
    bot = P.Scene(1920, 1080, tile_y0=536, tile_rows=544)
    parts = []
    for t in (top, bot):
        P.scenes.cornell(t)
        t.setPerfSamplesPerPixel(4)
        t.setMaxBounceDepth(4)
        t.initBlueNoise()
        t.uploadToGPU()
        parts.append(t.render_to_host())
        t.close()
    assert np.array_equal(np.concatenate([parts[1], parts[0]], axis=0), rgb)
    s.close()


def _frames(P, build, opts, W=1920, H=1080, spp=4, depth=4, n_frames=2):
    s = P.Scene(W, H)
    build(s)
    s.setPerfSamplesPerPixel(spp)
    s.setMaxBounceDepth(depth)
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    s.initBlueNoise()
    s.uploadToGPU()
    s.set_option("count_rays", 1)
    for k, v in opts.items():
        s.set_option(k, v)
    out = []
    for _ in range(n_frames):
        rgb = s.render_to_host()
        out.append(dict(rgb8=rgb, stats=s.stats(), render_mode=s.get_option("render_mode"), **{k: s.read(b) for k, b in (
            ("accum", P.BUF_ACCUM), ("normal", P.BUF_NORMAL), ("depth", P.BUF_DEPTH), ("object_id", P.BUF_OBJECT_ID),
            ("rng", P.BUF_RNG))}))
    s.close()
    return out


def _many(P, s):
    from test_parity_gpu import _many_meshes
    _many_meshes(P, s, n=60)


@pytest.mark.parametrize("scene", ["showcase", "fluid", "cornell", "many"])
def test_full_size_traversal_variants_agree(P, scene):
    """BASELINE's full size (1920x1080, 4 spp / 2 spp for the fluid scene, 4 bounces, 2 frames): the default
    traversal -- pair queue, compacted leaf phase, shadow-ray subtree stealing, early-yielding descent, split
    pair batches -- against the plain one (batches of 64 pairs, every lane its own leaf, no stealing) and
    against the lock-step mesh loop, the asynchronous-lane kernel and the wavefront stages: every buffer, the generator states and the ray counts are equal.  16 M rays
    per frame reach the rare cases (exact-t ties on shared edges, hits on leaf-box faces) that small frames
    in the oracle tests may not."""
    build = {"showcase": P.scenes.showcase, "cornell": P.scenes.cornell, "many": lambda s: _many(P, s),
             "fluid": lambda s: P.scenes.fluid(s, cells=256, t=0.3)}[scene]
    spp = 2 if scene == "fluid" else 4
    ref = _frames(P, build, {}, spp=spp)
    plain = dict(fetch_min=0, leaf_pairs=0, steal=0, csteal=0, leaf_min=64, pair_split=0)
    # ("many": 68 meshes behind a real TLAS -- PMODE 3 rounds against plain rounds and the lock-step general walk;
    #  the async / wavefront kernels take single-leaf TLASes only and fall back to the same default there)
    # merged=0: separate closest-hit and any-hit phases (PMODE 2) instead of one traversal per iteration (PMODE 4)
    # lds_nodes=1: four tiles per workgroup sharing an LDS copy of the mesh heads and of the BLAS top levels
    # pm1_wg=2: two tiles per workgroup sharing one LDS copy of a small scene, six waves per SIMD (PMODE 1)
    # sample_sync=0 / 1: a lane starts its next sample as soon as its path has ended / the lanes of a wave start their samples
    #   together (the default follows the depth limit and the mode; both are forced here for all four scenes)
    # refill=2: PMODE 1 as persistent waves whose lanes draw the next pixel of the launch (the default for overlapping frames)
    # tlas_rounds=1: shadow rays behind a real TLAS take one leaf per fill (the path of scenes with more than 1024 meshes)
    # csteal: closest-hit subtree stealing with verification (PMODE 2's default) off / at its most eager / without the thieves following their victims' limits
    # tile_run: the one-tile kernels' workgroup -> tile map (default: of every 64 tiles each XCD renders 8 neighbours): tile k on workgroup k / runs of 3
    for opts in (dict(tile_run=0), dict(tile_run=3), dict(merged=1), dict(merged=1, csteal=0), dict(merged=1, steal=1, csteal=1), dict(merged=0, csteal=0), dict(merged=0, csteal=1, csteal_leaf_min=4), dict(merged=0, csteal=3, csteal_min=4, csteal_follow=0), dict(lds_nodes=1), plain, dict(pair_trace=0), dict(async_lanes=1), dict(wavefront=1), dict(pm1_wg=2), dict(tlas_rounds=1), dict(refill=2), dict(sample_sync=0), dict(sample_sync=1), dict(sample_sync=1, refill=2)):
        got = _frames(P, build, opts, spp=spp)
        for f, (a, b) in enumerate(zip(ref, got)):
            for k in ("accum", "normal", "depth", "object_id", "rgb8", "rng"):
                av, bv = a[k], b[k]
                if av.dtype == np.float32:
                    av, bv = av.view(np.uint32), bv.view(np.uint32)
                assert np.array_equal(av, bv), f"{scene} {opts} frame {f}: {k} differs in {(av != bv).sum()} words"
            # (the asynchronous-lane kernel walks every shadow ray; the others count the zero-valued light samples apart)
            # (render_mode 2: it really ran -- a scene with a real TLAS falls back to the default kernel)
            want = dict(a["stats"], shadow_rays_walked=a["stats"]["shadow_rays"]) if b["render_mode"] == 2 else a["stats"]
            assert want == b["stats"], f"{scene} {opts} frame {f}"
            assert a["stats"]["shadow_rays_walked"] <= a["stats"]["shadow_rays"]
    assert ref[0]["accum"].any()


@pytest.mark.parametrize("scene,spp,frames", [("cornell", 4, 2), ("showcase", 4, 1)])
def test_full_size_frames_equal_the_oracle(P, O, blue_noise, scene, spp, frames):
    """BASELINE configs[1] and configs[2] at their FULL size (1920x1080, 4 spp, 4 bounces): every buffer, generator state and ray
    count of the GPU frame against the oracle's (16 host threads: about a second per Cornell frame, a few for the showcase)."""
    s = P.Scene(1920, 1080)
    getattr(P.scenes, scene)(s)
    gpu, cpu = render_both(P, O, s, blue_noise, spp, 4, frames, threads=16)
    assert_frames_equal(gpu, cpu)
    s.close()


@pytest.mark.parametrize("scene", ["cornell", "showcase", "cornell+post", "cornell+tm"])
def test_pipelined_frames_are_the_same_frames(P, scene):
    """Consecutive frames into ALTERNATING device targets overlap on the device (ptrt_set_option "pipeline", the default: a
    frame's launches follow the previous frame's launches of the same tile rows on auxiliary streams and do not wait for
    the stream the caller sees).  Every frame -- RGB8, HDR image, generator states, ray counts -- equals the frame rendered
    with the launches ordered behind the stream; an entry point that may touch device state (here: a generator reset, a
    camera is not one), the same target twice, or a handed-out buffer pointer make a frame wait again."""
    import torch
    W, H = 640, 360
    post = scene.endswith("+post")  # denoiser + bloom: the trace alternates between two sets of HDR image and G-buffers
    build = P.scenes.cornell if scene.startswith("cornell") else (lambda s: P.scenes.showcase(s, segments=16))

    def run(pipeline):
        s = P.Scene(W, H)
        build(s)
        s.setPerfSamplesPerPixel(4 if scene in ("cornell", "cornell+tm") else 2)  # (4 x 4: the overlapping Cornell frames run with lane refill)
        s.setMaxBounceDepth(4)
        s.setDenoiserEnabled(post)
        s.setBloomEnabled(post)
        s.initBlueNoise()
        s.uploadToGPU()
        s.set_option("count_rays", 1)
        s.set_option("merged", 0)  # (no loop-shape sampling: it orders its frames behind the stream)
        s.set_option("pipeline", pipeline)
        s.set_option("persist", 2)  # (512 persistent waves: this small frame then has enough tiles per wave for lane refill)
        if scene == "cornell+tm":  # the refill kernel's tonemap pass on a stream of the highest priority, its waves at s_setprio 3,
            s.set_option("tm_prio", 3)      # and events around every launch (ptrt_launch_ms_history): same frames
            s.set_option("time_launches", 1)
        tgt = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
        frames, flags = [], []
        for f in range(8):
            if f == 5:
                s.moveCamera((0.3, 0.1, 5.0))  # host state only: the next frame still overlaps
            s.render_to_device(tgt[f & 1].data_ptr())
            flags.append(s.get_option("pipelined"))
            assert s.get_option("refilled") == (1 if scene in ("cornell", "cornell+tm") and flags[-1] else 0)
            if f in (2, 7):  # (reading back in between would order everything: only here)
                s.sync()
                frames.append((tgt[f & 1].cpu().numpy().copy(), s.read(P.BUF_ACCUM), s.read(P.BUF_RNG), s.stats()))
        # the same target twice: the second frame waits for the stream
        s.render_to_device(tgt[1].data_ptr())
        s.render_to_device(tgt[1].data_ptr())
        same_target = s.get_option("pipelined")
        s.reset_rng(P.DEFAULT_SEED)
        s.render_to_device(tgt[0].data_ptr())
        after_reset = s.get_option("pipelined")
        s.close()
        return frames, flags, same_target, after_reset

    on, flags_on, same_on, reset_on = run(1)
    off, flags_off, _, _ = run(0)
    assert flags_off == [0] * 8 and same_on == 0 and reset_on == 0
    # frame 0 waits (uploads came before it); from the second alternation on the frames overlap -- the read-back after
    # frame 2 (a synchronising entry point) makes frame 3 wait once more
    assert flags_on[0] == 0 and flags_on[1] == 1 and flags_on[2] == 1 and flags_on[3] == 0 and flags_on[4:] == [1, 1, 1, 1], flags_on
    for (a_rgb, a_acc, a_rng, a_st), (b_rgb, b_acc, b_rng, b_st) in zip(on, off):
        assert np.array_equal(a_rgb, b_rgb) and np.array_equal(a_acc.view(np.uint32), b_acc.view(np.uint32))
        assert np.array_equal(a_rng, b_rng) and a_st == b_st
    assert on[1][0].any()


def test_post_chain_switched_off_between_overlapping_frames(P):
    """a frame WITHOUT a post chain must not overlap a predecessor WITH one -- that frame's denoiser / bloom still
    reads the HDR image and G-buffers the trace would overwrite (the application toggles perfSettings.enableBloom / enableDenoiser,
    which reach the back end through host-only entry points).  Such a frame waits for the stream; the frames before and after
    equal the ones rendered with pipeline = 0."""
    import torch
    W, H = 640, 360

    def run(pipeline):
        s = P.Scene(W, H)
        P.scenes.cornell(s)
        s.setPerfSamplesPerPixel(2)
        s.setMaxBounceDepth(3)
        s.setDenoiserEnabled(True)
        s.setBloomEnabled(True)
        s.initBlueNoise()
        s.uploadToGPU()
        s.set_option("pipeline", pipeline)
        tgt = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
        flags, frames = [], []
        for f in range(9):
            if f == 4:  # the application switches the chain off ...
                s.setDenoiserEnabled(False)
                s.setBloomEnabled(False)
            if f == 7:  # ... and on again
                s.setDenoiserEnabled(True)
                s.setBloomEnabled(True)
            s.render_to_device(tgt[f & 1].data_ptr())
            flags.append(s.get_option("pipelined"))
            if f in (3, 4, 5, 8):
                torch.cuda.synchronize()
                frames.append(tgt[f & 1].cpu().numpy().copy())
        s.sync()
        frames.append(s.read(P.BUF_ACCUM))
        frames.append(s.read(P.BUF_RNG))
        s.close()
        return flags, frames

    flags, on = run(1)
    _, off = run(0)
    # frame 4 (the first without a chain) waits for frame 3's chain; 5 and 6 overlap again; a frame WITH a chain may follow one
    # without (it writes the other set and nothing reads the current one behind the fused tonemap)
    assert flags[1:4] == [1, 1, 1] and flags[4] == 0 and flags[5:7] == [1, 1] and flags[7] == 1, flags
    for a, b in zip(on, off):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_render_recorded_into_a_graph_then_rendered_directly(P):
    """while the caller records the stream into a hipGraph, ptrt_render issues ONE ordered launch and records none
    of its pipelining bookkeeping events into the caller's graph; the first frame after the recording waits for the stream
    (a replay never passes through ptrt_render).  The replayed frame and the direct frame after it are the frames of a plain run."""
    import torch
    W, H = 320, 200

    def scene():
        s = P.Scene(W, H)
        P.scenes.cornell(s)
        s.setPerfSamplesPerPixel(2)
        s.setMaxBounceDepth(3)
        s.setDenoiserEnabled(False)
        s.setBloomEnabled(False)
        s.initBlueNoise()
        s.uploadToGPU()
        return s

    ref = scene()
    plain = []
    t = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    for f in range(3):
        ref.render_to_device(t.data_ptr())
        torch.cuda.synchronize()
        plain.append(t.cpu().numpy().copy())
    ref_rng = ref.read(P.BUF_RNG)
    ref.close()

    s = scene()
    st = torch.cuda.Stream()
    s.set_stream(st.cuda_stream)
    tgt = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    s.render_to_device(tgt[0].data_ptr())  # frame 0, direct
    st.synchronize()
    assert np.array_equal(tgt[0].cpu().numpy(), plain[0])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
        s.render_to_device(tgt[1].data_ptr())  # frame 1: recorded, not run
        assert s.get_option("pipelined") == 0 and s.get_option("split_eff") == 1
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(tgt[1].cpu().numpy(), plain[1])
    s.render_to_device(tgt[0].data_ptr())  # frame 2, direct, into the other target: must not overlap anything
    assert s.get_option("pipelined") == 0
    s.render_to_device(tgt[1].data_ptr())  # ... and from here on frames overlap again
    assert s.get_option("pipelined") == 1
    st.synchronize()
    s.set_stream(0)
    del g
    # (frame 2 went into tgt[0]; frame 3 into tgt[1])
    assert np.array_equal(tgt[0].cpu().numpy(), plain[2])
    s.close()
    assert ref_rng.shape[0] == W * H


def test_launch_durations_of_overlapping_frames(P):
    """ptrt_launch_ms_history (ABI 6): with option time_launches the launches of frames that overlap carry their own events, on
    the auxiliary stream each runs on -- `split` entries per frame, the trace kernel's duration and (lane refill) the tonemap
    pass behind it; frames that are one launch on the stream have none."""
    import torch
    W, H = 1920, 1080
    s = P.Scene(W, H)
    P.scenes.cornell(s)
    s.setPerfSamplesPerPixel(4)
    s.setMaxBounceDepth(4)
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    s.initBlueNoise()
    s.uploadToGPU()
    tgt = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    for f in range(3):
        s.render_to_device(tgt[f & 1].data_ptr())
    a, b = s.launch_ms_history()
    assert len(a) == 0  # (option off)
    s.set_option("time_launches", 1)
    for f in range(3, 10):
        s.render_to_device(tgt[f & 1].data_ptr())
        # (reading the history synchronises and counts as touching the context: the frame behind it waits for the stream)
        assert s.get_option("pipelined") == (0 if f == 3 else 1) and s.get_option("refilled") == (0 if f == 3 else 1)
    a, b = s.launch_ms_history()
    assert len(a) == 12 and len(b) == 12 and (a > 0.05).all() and (a < 50).all() and (b > 0).all()
    k = s.kernel_ms_history(6)
    assert len(k) == 6
    s.set_option("pipeline", 0)
    s.render_to_device(tgt[1].data_ptr())
    a, b = s.launch_ms_history()
    assert len(a) == 0  # (the run of timed overlapping frames ended)
    s.close()


def test_lane_refill_is_chosen_where_it_pays(P):
    """ptrt_render's rule for the lane-refill kernel (option "refill" = 1, the default): PMODE 1 frames that overlap
    their predecessor, with the simple materials, no post chain, at least 16 sample-bounces per pixel and at least two tiles
    per persistent wave -- the BASELINE headline at its full size qualifies from its second frame on, none of the others do."""
    import torch

    def refilled(W, H, spp, depth, frames=3, post=False, scene="cornell", same_target=False):
        s = P.Scene(W, H)
        (P.scenes.cornell if scene == "cornell" else (lambda sc: P.scenes.showcase(sc, segments=8)))(s)
        s.setPerfSamplesPerPixel(spp)
        s.setMaxBounceDepth(depth)
        s.setDenoiserEnabled(post)
        s.setBloomEnabled(False)
        s.initBlueNoise()
        s.uploadToGPU()
        s.set_option("merged", 0)
        tgt = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
        out = []
        for f in range(frames):
            s.render_to_device(tgt[0 if same_target else f & 1].data_ptr())
            out.append(s.get_option("refilled"))
        s.close()
        return out

    assert refilled(1920, 1080, 4, 4) == [0, 1, 1]                      # the headline: every frame that overlaps
    assert refilled(1920, 1080, 4, 4, same_target=True) == [0, 0, 0]    # one target: frames wait for the stream
    assert refilled(1920, 1080, 1, 4) == [0, 0, 0]                      # short pixels
    assert refilled(1920, 1080, 4, 4, post=True) == [0, 0, 0]           # a post chain beside the trace
    assert refilled(640, 360, 4, 4) == [0, 0, 0]                        # too few tiles per persistent wave
    assert refilled(1920, 1080, 4, 4, scene="showcase") == [0, 0, 0]    # trees: not PMODE 1


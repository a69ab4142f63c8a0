// ptrt_capi.hip -- implementation of the C ABI in include/ptrt.h over HIP (gfx950).
//
// Host responsibilities here: own every device allocation of a context, re-lay-out
// the caller's reference-shaped scene arrays (40-byte nodes, index triples,
// per-mesh pointers; mesh.cuh:37-47, intersection.cuh:90-106) into the arena the
// kernels read (pt_kernels.hip.h), pick the kernel instantiation that matches the
// scene, launch on the context's stream, time kernels with HIP events.
// There is no CPU rendering path in this library.
#include "../../include/ptrt.h"
#include "pt_build.hip.h"
#include "pt_denoise.hip.h"
#include "pt_post.hip.h"
#include "pt_refit.hip.h"
#include "pt_render.hip.h"
#include "pt_wavefront.hip.h"
#include "pt_async.hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <climits>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <algorithm>
#include <vector>

namespace {

constexpr int EV_RING = 256;
// loop-shape choice of the queue modes (ptrt_set_option "merged" = -1): frames that only warm the clocks up, samples per
// shape (odd: their median decides), and the gain the merged loop must show to replace the separate-phase default
constexpr int TUNE_WARM = 4, TUNE_SAMPLES = 3;
constexpr float TUNE_MIN_GAIN = 0.005f;
thread_local std::string g_last_error = "";
std::mutex g_live_mutex;
std::set<ptrt_ctx *> g_live;

} // namespace

struct ptrt_ctx {
    int device = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;
    std::vector<hipEvent_t> ev_ring; // 2 * EV_RING events: start/stop per launch
    unsigned long long launches = 0;
    int W = 0, H = 0, y0 = 0, rows = 0;
    int il_period = 1, il_phase = 0; // > 1: rows = the 8-row strips il_phase, il_phase + il_period, ... of the frame
    size_t npix = 0;
    std::string err;

    // frame buffers (tile)
    uint32_t *d_rng = nullptr;
    float *d_accum = nullptr, *d_normal = nullptr, *d_depth = nullptr;
    int *d_object_id = nullptr;
    // second set of the four, allocated when frames with a post chain overlap (ptrt_render "pipeline"): the trace of frame
    // N + 1 writes one set while the denoiser / bloom of frame N still reads the other; d_* is always the latest frame's
    float *alt_accum = nullptr, *alt_normal = nullptr, *alt_depth = nullptr;
    int *alt_object_id = nullptr;
    unsigned char *d_rgb8 = nullptr;
    unsigned char *last_rgb8 = nullptr; // where the last frame's RGB8 went
    void *last_frame_target = nullptr;  // ... or the caller's frame it was written into (PTRT_OUT_DEVICE_FRAME)
    int time_kernels = 1;               // option: record the two events per launch that ptrt_kernel_ms_history reads
    unsigned long long *d_counters = nullptr; // n_counter_slots x pt::COUNTER_WORDS {extension rays, shadow rays, paths, zero-valued light samples}
    size_t n_counter_slots = 0;
    float2 *d_blue = nullptr;
    uint32_t *d_jump = nullptr;
    int n_jump = 0;
    bool rng_ready = false;

    // scene arena
    float4 *d_nodes2 = nullptr; // two-level records derived from d_nodes (pt::expand_nodes_kernel), NODE2_F4 float4 per node
    float4 *d_mesh_recs = nullptr, *d_nodes = nullptr, *d_tris = nullptr, *d_tlas_nodes = nullptr,
           *d_materials = nullptr, *d_lights = nullptr;
    int2 *d_leaves = nullptr, *d_tlas_leaves = nullptr;
    int *d_tlas_mesh_ids = nullptr;
    float4 *d_tlas_heads = nullptr; // mesh-record heads in TLAS-leaf order (PMODE 3), gathered before each frame
    float4 *d_inst_pre = nullptr;   // per mesh: world-space first-pass box of an instance (PMODE 3), host-computed
    float inst_c2 = 0.0f;           // ... and the per-ray growth factor that goes with it
    bool inst_pre_ok = false;       // false once device-side boxes have moved (GPU refit / rebuild) since it was computed
    int n_tlas_index = 0;
    std::vector<float4> h_mesh_recs;
    std::vector<unsigned char> h_shadow_skip; // per material: transmission > 0.5
    int n_meshes = 0, n_materials = 0, n_lights = 0;
    float4 *d_tlas_root_box = nullptr; // {bmin, bmax}
    int tlas_root_ref = 0;
    // refit support (ptrt_update_vertices / ptrt_refit)
    float *d_verts = nullptr;       // all meshes' vertices, xyz packed
    int4 *d_slot_face = nullptr;    // per leaf slot: global vertex indices + face index
    int *d_leaf_dst = nullptr;      // per leaf: where its box is stored (see convert_tree)
    int *d_node_dst = nullptr;      // per inner node: where its own box is stored
    int *d_level_nodes = nullptr;   // inner nodes ordered by depth
    std::vector<int> level_offset;  // level_offset[d] .. level_offset[d+1]: nodes at depth d+1
    std::vector<int> mesh_vert_base, mesh_vert_count;
    int n_slots = 0, n_leaves = 0;
    // GPU rebuild support (ptrt_build_bvh, pt_build.hip.h)
    int4 *d_face_src = nullptr;   // per face, mesh-major, original order: global vertex indices + face index
    int *d_slot_pos = nullptr;    // per leaf slot: its position in the mesh's prim order
    std::vector<int> mesh_face_base, mesh_face_count, mesh_slot_base, mesh_slot_count;
    std::vector<unsigned char> mesh_rebuildable, mesh_is_soup;
    uint32_t *d_sort_keys[2] = {nullptr, nullptr}, *d_sort_vals[2] = {nullptr, nullptr}, *d_sort_hist = nullptr,
             *d_cbounds = nullptr;
    float *d_centroids = nullptr;
    int sort_capacity = 0;
    std::map<int, hipGraphExec_t> graphs; // refit (-1) / rebuild-of-mesh-m (m) launch sequences
    hipStream_t capture_stream = nullptr;
    int use_graphs = 0; // measured: replaying the sequence as a hipGraph gains nothing on ROCm 7.2
    bool tlas_single_leaf = false, all_single_leaf = false, mats_full = false;
    int stack_entries = 1;
    int pair_meshes = 0, pair_tri_slots = 0, pair_max_leaf = 0;
    int tlas_max_leaf = 0, tlas_depth = 0;
    bool have_geometry = false, have_materials = false;

    pt::Camera cam{};
    pt::f3 sky_top{0.6f, 0.7f, 1.0f}, sky_bottom{1.0f, 1.0f, 1.0f};
    int use_sky = 1;
    float4 *d_env = nullptr; // equirectangular environment map (ptrt_set_env_map)
    int env_w = 0, env_h = 0;

    // denoiser (class Denoiser, denoiser.cuh:781-1070): scratch + double-buffered history
    bool dn_on = false, dn_first = true;
    pt::DenoiseSettings dn{};
    // float4 images (pt_denoise.hip.h): cur4 {rgb,0}; c4 {rgb,variance} ping-pong; g4 {normal,depth},
    // h1 {mean,len}, h2 {m2,-} double-buffered across frames
    float4 *dn_cur4 = nullptr, *dn_c4[2] = {nullptr, nullptr}, *dn_g4[2] = {nullptr, nullptr};
    float4 *dn_h1[2] = {nullptr, nullptr}, *dn_h2[2] = {nullptr, nullptr};
    int *dn_hobj = nullptr;
    float *dn_motion = nullptr, *dn_out = nullptr, *dn_pvp = nullptr;
    int dn_cur = 0; // which history set holds the latest result
    int dn_active = 1, mv_active = 1;
    float prev_view_proj[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

    // post chain (pt_post.hip.h): render size (perfSettings.resolutionScale) and bloom
    int rw = 0, rh = 0; // render size; == W,H unless ptrt_set_render_size asked for less
    float *s_accum = nullptr, *s_normal = nullptr, *s_depth = nullptr; // the d_scaled_* set (scene.cuh:181-186)
    int *s_object_id = nullptr;
    int bloom_on = 0;
    float *bl_mip[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // d_bloom_mip_chain (scene.cuh:161, 809-822)
    bool scaled() const { return rw != W || rh != H; }
    size_t rpix() const { return scaled() ? (size_t)rw * rh : npix; } // pixels the path tracer renders

    // presentation ring (ptrt_present_*): device RGB8 frames mirrored into pinned host memory
    struct PresentSlot {
        unsigned char *dev = nullptr, *host = nullptr;
        hipEvent_t rendered = nullptr, done = nullptr;
        bool in_flight = false;
    };
    std::vector<PresentSlot> present;
    hipStream_t present_stream = nullptr;

    // wavefront stages (pt_wavefront.hip.h): per-path state, planar, tile-ordered
    int wavefront = 0; // option: 1 = render scenes with a single-leaf TLAS through the trace/shade stages
    int wf_sort = 0;   // option: the shade stage bins its paths by class first (active-path sorting, pt_wavefront.hip.h)
    uint32_t *wf_st = nullptr, *wf_occ = nullptr, *wf_live = nullptr;
    float *wf_planes = nullptr; // 25 planes: ray 6, thr 3, acc 3, avg 3, pend 3, shadow ray 7
    float4 *wf_hit = nullptr;
    size_t wf_items = 0;
    int wf_trace_blocks = 0;
    size_t wf_trace_lds = 0;
    int n_geometry_uploads = 0, n_instance_updates = 0;
    bool any_transform = false; // some mesh is an instance with its own transform
    int pair_split = 1;         // option (A/B, tests)
    int last_mode = 0; // how the last frame was rendered: 0 megakernel, 1 wavefront stages, 2 asynchronous lanes
    // asynchronous-lane megakernel (pt_async.hip.h)
    int async_lanes = 0, shade_min = 32, as_leaf_min = 24; // options
    uint32_t *as_cursor = nullptr;
    // Lane refill (PMODE 1, path_trace_kernel<.., STREAM = true>): option "refill" 0 never, 1 (default) where it was measured to
    // pay -- frames that overlap their predecessor, simple materials, no post chain, spp * bounces >= 16 --, 2 wherever PMODE 1 runs (tests)
    int refill = 1;
    bool refill_eff = false;         // ... the last frame
    static constexpr int STAGES = 4;
    void *h_stage[STAGES] = {nullptr, nullptr, nullptr, nullptr}; // ptrt_update_vertices from host memory: pinned staging, in rotation
    size_t stage_bytes[STAGES] = {0, 0, 0, 0};
    hipEvent_t stage_ev[STAGES] = {nullptr, nullptr, nullptr, nullptr};
    unsigned long long stage_n = 0;
    float *d_stage[STAGES] = {nullptr, nullptr, nullptr, nullptr}; // ... from PINNED host memory: device staging behind a copy stream
    size_t d_stage_bytes[STAGES] = {0, 0, 0, 0};
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_ev = nullptr;
    unsigned int *d_queue = nullptr; // {ticket, waves out} per launch lane: [0] the stream, [1 + i] auxiliary stream i
    int sample_sync = -1;            // option "sample_sync": -1 (default) where it was measured to pay, 0 never, 1 always (ptrt_render)
    int sample_sync_eff = 0;         // ... the last frame
    int tile_run = 8;                // option "tile_run": of every 8 * run consecutive tiles XCD x renders a run of neighbours (path_trace_kernel); 0: tile k on workgroup k
    int ticket_tiles = 1;            // option "ticket_tiles": consecutive tiles per ticket of the queue
    int persist = 0, n_cus = 0;      // option "persist": persistent waves per CU (0 = the variant's occupancy)
    int as_blocks[2] = {0, 0}; // resident workgroups of the <false>/<true> kernel at as_lds bytes of LDS
    size_t as_lds = 0;

    // options
    int count_rays = 0, force_geom = -1, force_full = 0, pair_trace = 1, fetch_min = 16, leaf_pairs = 1, steal = 1, leaf_min = 8;
    int atrous_exp = 0; // option: 1 = the a-trous luminance weight through v_exp_f32 (the reference's __expf) instead of det_exp: tolerance mode
    int csteal_follow = 1, csteal_leaf_min = 32;
    int csteal = 2, csteal_min = 0; // options: PMODE 2 closest-hit subtree stealing with verification (pt_render.hip.h run_closest_queue)
    int lds_pad = 0; // extra bytes of LDS per workgroup (A/B of the occupancy)
    // PMODE 1, simple materials: tiles per workgroup.  1 (default): five waves per SIMD.  2: two tiles share the LDS copies, six
    // waves per SIMD on 80 VGPRs -- measured on Cornell 1080p: 1.875 vs 1.877 ms, the 112 B per lane it spills eat what 24
    // instead of 20 waves per CU bring.  0: two when the scene fits that budget.
    int pm1_wg = 1;
    // option "split": the frame's tile rows dealt to that many launches on auxiliary streams of the context (forked from and
    // joined to its stream by events, so the caller still sees one stream).  Concurrent launches of ONE frame buy nothing
    // (Cornell 1080p 1.85 vs 1.82 ms); what they make possible is option "pipeline": frame N + 1's launches follow frame N's
    // on their own streams and do not wait for the rest of frame N to drain -- 1.81 -> 1.66 ms (see ptrt_render).
    static constexpr int MAX_SPLIT = 4;
    int split = 2, split_eff = 1, pipeline = 1;
    hipStream_t aux_stream[MAX_SPLIT] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t split_fork = nullptr, split_join[MAX_SPLIT] = {nullptr, nullptr, nullptr, nullptr};
    bool heads_fresh = false;     // PMODE 3: the TLAS-leaf-order heads were gathered since the last touching entry point
    bool touched = true;          // an entry point that may enqueue device work ran since the last ptrt_render (ctx_live)
    bool escaped = false;         // ptrt_device_buffer handed out pointers into the context's buffers
    void *prev_out = nullptr;     // the previous frame's device target
    hipStream_t prev_stream = nullptr;
    int prev_split = 0;
    bool pipelined_last = false;  // the last frame's launches did not wait for the stream (ptrt_get_option "pipelined")
    hipEvent_t head_ev[2] = {nullptr, nullptr}; // the stream's head at the start of the last two ptrt_render calls
    unsigned head_n = 0;
    int tlas_rounds = 0; // option (A/B, tests): PMODE 3 shadow rays take one TLAS leaf per fill, as scenes with > 1024 meshes do
    int stage = 7; // PMODE 1, shading inputs staged in LDS: 0 none, else jitter table + blue noise, | 1 lights, | 2 materials
    int lds_nodes = 0; // option: PMODE 2 in 256-thread workgroups sharing an LDS copy of the BLAS top levels (measured slower: DESIGN.md 3.1)
    int n_nodes = 0;
    // option: PMODE 4 (one traversal per loop iteration) where PMODE 2 applies: 0 / 1, or -1 = try both on a scene's first
    // frames and keep the faster (showcase: PMODE 4 by 1.5 %; fluid, 1 M triangles: PMODE 2 by 2-5 %; the bits are the same)
    int merged = -1;
    int merged_eff = 0;                  // what this launch uses
    int tune_n = 0, tune_choice = -1;    // auto: frames measured so far (variants alternate), the decision (-1: none yet)
    unsigned long long tune_key = ~0ull, tune_launch[2 * TUNE_SAMPLES] = {};
    int last_pmode = 0;                  // PMODE of the last megakernel launch (ptrt_get_option "pmode")
    bool last_merged_possible = false;   // ... and whether that scene has the two loop shapes to choose from
    bool timed = false;
    bool prev_post = false;              // the previous frame had a denoiser / bloom chain behind its trace (ptrt_render "pipeline")
    size_t refill_counter_base = 0;      // first counter slot of the lane-refill kernel's launches (disjoint from the tile slots)
    // option "time_launches": three events per launch of a frame dealt to the auxiliary streams -- before the trace kernel,
    // behind it, behind the tonemap pass that follows it with lane refill -- on the stream the launch runs on, so that a
    // measurement can state the duration of the very launches it timed (ptrt_launch_ms_history); the events around a frame on
    // the context's stream (time_kernels) measure the frame INTERVAL once frames overlap
    int time_launches = 0;
    std::vector<hipEvent_t> launch_ev[MAX_SPLIT]; // 3 * EV_RING per auxiliary stream
    unsigned char launch_timed[EV_RING] = {};     // launches of frame (launches % EV_RING) that carry events (0: none)
    // option "tm_prio" (lane refill's tonemap pass): bit 0 = the pass runs on a stream of the highest priority, forked from
    // and joined to the launch's stream by events; bit 1 = its waves raise their issue priority (s_setprio 3)
    int tm_prio = 0;
    hipStream_t tm_stream[MAX_SPLIT] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t tm_fork[MAX_SPLIT] = {nullptr, nullptr, nullptr, nullptr}, tm_join[MAX_SPLIT] = {nullptr, nullptr, nullptr, nullptr};
};

// Presentation ring without a context (ptrt_ring_*): the CUDA-registered GL pixel-buffer object of
// rtgl::init_interop_viewer as `slots` device frames mirrored into pinned host memory.
struct ptrt_ring {
    int device = 0;
    size_t bytes = 0;
    struct Slot {
        unsigned char *dev = nullptr, *host = nullptr;
        hipEvent_t rendered = nullptr, done = nullptr;
        bool in_flight = false; // a download of this slot has been enqueued and not yet waited for
        bool marked = false;    // `rendered` was recorded by the render call that wrote the slot
    };
    std::vector<Slot> slots;
    hipStream_t copy_stream = nullptr;
};

namespace {

std::mutex g_ring_mutex;
std::set<ptrt_ring *> g_rings;

// ptrt_render / ptrt_post_frame wrote their RGB8 frame to `out` on `stream`: if that is a ring slot, the slot's
// download must wait for exactly this point of the stream.
void ring_mark_rendered(const void *out, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_ring_mutex);
    for (ptrt_ring *r : g_rings)
        for (auto &s : r->slots)
            if (s.dev == out) {
                s.marked = hipEventRecord(s.rendered, stream) == hipSuccess;
                return;
            }
}

// Records the message for ptrt_last_error.  `c` may be a stale (already destroyed) handle -- every entry point
// reports "bad context" through here -- so it is only written to while it is in the live set.
int fail(ptrt_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (c) {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        if (g_live.count(c))
            c->err = buf;
    }
    return code;
}

#define HIP_TRY(c, call)                                                                                       \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail((c), PTRT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));                       \
    } while (0)

template <class T> void dfree(T *&p) {
    if (p) {
        (void)hipFree(p);
        p = nullptr;
    }
}
template <class T> int upload(ptrt_ctx *c, T *&dst, const std::vector<T> &src) {
    dfree(dst);
    const size_t n = src.empty() ? 1 : src.size();
    HIP_TRY(c, hipMalloc((void **)&dst, n * sizeof(T)));
    if (!src.empty())
        HIP_TRY(c, hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // src may be a temporary
    return PTRT_OK;
}

float4 f4(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
float as_f(int i) {
    float f;
    std::memcpy(&f, &i, 4);
    return f;
}

// ---- GF(2) algebra for the XORWOW subsequence jump (cuRAND: subsequence = 2^67 draws) ----
struct GF2 {
    uint32_t col[160][5];
};
void gf2_apply(const GF2 &m, const uint32_t in[5], uint32_t out[5]) {
    uint32_t a[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 5; ++w)
        for (uint32_t bits = in[w]; bits; bits &= bits - 1) {
            const uint32_t *c = m.col[w * 32 + __builtin_ctz(bits)];
            for (int k = 0; k < 5; ++k)
                a[k] ^= c[k];
        }
    std::memcpy(out, a, sizeof a);
}
void gf2_square(GF2 &m) {
    GF2 t;
    for (int j = 0; j < 160; ++j)
        gf2_apply(m, m.col[j], t.col[j]);
    m = t;
}
// jump[k] = (one generator step)^(2^(67+k)), k = 0..JUMP_MAX-1, as 800 words each.  Computed once for the largest
// pixel index a context can have (2^40 pixels), so the vector never reallocates under a concurrent reader.
constexpr int JUMP_MAX = 40;
const std::vector<uint32_t> &jump_matrices() {
    static std::vector<uint32_t> out;
    static std::once_flag once;
    std::call_once(once, [] {
    const int n = JUMP_MAX;
    GF2 m;
    for (int j = 0; j < 160; ++j) { // image of basis vector j under one step of the recurrence
        uint32_t v[5] = {0, 0, 0, 0, 0};
        v[j / 32] = 1u << (j % 32);
        const uint32_t t = v[0] ^ (v[0] >> 2);
        uint32_t nv[5] = {v[1], v[2], v[3], v[4], (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1))};
        std::memcpy(m.col[j], nv, sizeof nv);
    }
    for (int i = 0; i < 67; ++i)
        gf2_square(m);
    out.resize((size_t)n * 800);
    for (int k = 0; k < n; ++k) {
        std::memcpy(&out[(size_t)k * 800], m.col, 800 * sizeof(uint32_t));
        gf2_square(m);
    }
    });
    return out;
}

// ---- scene re-layout ---------------------------------------------------------------
struct Relayout {
    std::vector<float4> nodes; // child-pair inner nodes
    std::vector<int2> leaves;
    std::vector<float4> tris;
    int max_depth = 0;
};

// Converts one reference-shaped tree (pre-order 40-byte nodes) into child-pair nodes.
// `emit_leaf(start,count)` returns the leaf id for a leaf node.  Returns the root
// reference, or INT32_MIN on malformed input.
// `dst` codes say where a subtree's box is stored, for the GPU refit: >= 0 -> inner node
// (code >> 1), side (code & 1); < 0 -> root box of tree `root_dst`.  `emit_leaf(start,count,dst)`.
// `node_dst`/`node_depth` (optional) receive that code and the depth of every inner node created.
template <class EmitLeaf>
int convert_tree(const ptrt_bvh_node *in, int n_in, std::vector<float4> &out_nodes, EmitLeaf emit_leaf, int &max_depth,
                 std::string &why, int root_dst = -1, std::vector<int> *node_dst = nullptr,
                 std::vector<int> *node_depth = nullptr) {
    struct Item {
        int old_idx, new_idx, depth;
    };
    if (n_in <= 0) {
        why = "empty node array";
        return INT32_MIN;
    }
    std::vector<char> seen((size_t)n_in, 0);
    auto ref_of = [&](int old_idx, int depth, int dst, std::vector<Item> &work) -> int {
        if (old_idx < 0)
            return ~emit_leaf(0, 0, dst); // absent child: an empty leaf behind an unhittable box
        if (old_idx >= n_in) {
            why = "child index out of range";
            return INT32_MIN;
        }
        if (seen[old_idx]) {
            why = "node referenced twice (not a tree)";
            return INT32_MIN;
        }
        seen[old_idx] = 1;
        const ptrt_bvh_node &N = in[old_idx];
        if (N.count > 0)
            return ~emit_leaf(N.start, N.count, dst);
        const int ni = (int)(out_nodes.size() / 4);
        out_nodes.resize(out_nodes.size() + 4);
        if (node_dst) {
            node_dst->resize((size_t)ni + 1, 0);
            node_depth->resize((size_t)ni + 1, 0);
            (*node_dst)[ni] = dst;
            (*node_depth)[ni] = depth + 1;
        }
        work.push_back({old_idx, ni, depth + 1});
        return ni;
    };
    std::vector<Item> work;
    const int root = ref_of(0, 0, root_dst, work);
    if (root == INT32_MIN)
        return root;
    // Numbering: the top TOP_LEVELS levels in level order (a tree's first 2^TOP_LEVELS - 1 inner nodes are then its top
    // levels, which the LDS-staged variant of the trace kernel copies per workgroup), everything below depth first.
    size_t head = 0;
    while (head < work.size()) {
        Item it;
        if (work[head].depth <= pt::TOP_LEVELS) { // (a node of the top levels: first in, first out)
            it = work[head];
            ++head;
        } else {
            it = work.back();
            work.pop_back();
        }
        if (it.depth > max_depth)
            max_depth = it.depth;
        const ptrt_bvh_node &N = in[it.old_idx];
        const int L = ref_of(N.left, it.depth, it.new_idx * 2, work);
        if (L == INT32_MIN)
            return L;
        const int R = ref_of(N.right, it.depth, it.new_idx * 2 + 1, work);
        if (R == INT32_MIN)
            return R;
        const float BIG = 1e30f;
        ptrt_vec3 lmin{BIG, BIG, BIG}, lmax{-BIG, -BIG, -BIG}, rmin = lmin, rmax = lmax;
        if (N.left >= 0) {
            lmin = in[N.left].bmin;
            lmax = in[N.left].bmax;
        }
        if (N.right >= 0) {
            rmin = in[N.right].bmin;
            rmax = in[N.right].bmax;
        }
        float4 *o = &out_nodes[(size_t)it.new_idx * 4];
        o[0] = f4(lmin.x, lmin.y, lmin.z, lmax.x);
        o[1] = f4(lmax.y, lmax.z, rmin.x, rmin.y);
        o[2] = f4(rmin.z, rmax.x, rmax.y, rmax.z);
        o[3] = f4(as_f(L), as_f(R), 0.0f, 0.0f);
    }
    return root;
}

// TLAS part of an upload: converts the reference-shaped TLAS, uploads it and derives what the launch needs of it
int upload_tlas(ptrt_ctx *c, int mesh_count, const ptrt_bvh_node *tlas_nodes, int tlas_node_count,
                const int32_t *tlas_mesh_indices, int tlas_index_count, bool dry_run) {
    std::vector<float4> tnodes;
    std::vector<int2> tleaves;
    bool bad = false;
    std::string why;
    auto emit_tleaf = [&](int start, int count, int) -> int {
        if (count > 0 && (start < 0 || start + count > tlas_index_count)) {
            bad = true;
            count = 0;
        }
        tleaves.push_back(make_int2(start < 0 ? 0 : start, count));
        return (int)tleaves.size() - 1;
    };
    int tdepth = 0;
    const int troot = convert_tree(tlas_nodes, tlas_node_count, tnodes, emit_tleaf, tdepth, why);
    if (troot == INT32_MIN || bad)
        return fail(c, PTRT_E_INVALID, "malformed TLAS (%s)", bad ? "leaf range out of bounds" : why.c_str());
    if (tdepth > 23)
        return fail(c, PTRT_E_INVALID, "TLAS is %d levels deep; needs <= 23", tdepth);
    std::vector<int> tids(tlas_mesh_indices, tlas_mesh_indices + tlas_index_count);
    for (int id : tids)
        if (id < 0 || id >= mesh_count)
            return fail(c, PTRT_E_INVALID, "TLAS references mesh %d of %d", id, mesh_count);
    if (dry_run)
        return PTRT_OK;
    std::vector<float4> rootbox = {f4(tlas_nodes[0].bmin.x, tlas_nodes[0].bmin.y, tlas_nodes[0].bmin.z, 0.0f),
                                   f4(tlas_nodes[0].bmax.x, tlas_nodes[0].bmax.y, tlas_nodes[0].bmax.z, 0.0f)};
    if (int rc = upload(c, c->d_tlas_root_box, rootbox))
        return rc;
    if (int rc = upload(c, c->d_tlas_nodes, tnodes))
        return rc;
    if (int rc = upload(c, c->d_tlas_leaves, tleaves))
        return rc;
    if (int rc = upload(c, c->d_tlas_mesh_ids, tids))
        return rc;
    dfree(c->d_tlas_heads);
    HIP_TRY(c, hipMalloc((void **)&c->d_tlas_heads, (size_t)tlas_index_count * pt::TLAS_HEAD_F4 * sizeof(float4)));
    c->n_tlas_index = tlas_index_count;
    c->tlas_root_ref = troot;
    c->tlas_single_leaf = troot < 0;
    c->pair_meshes = troot < 0 ? tleaves[~troot].y : 0;
    c->tlas_depth = tdepth;
    c->tlas_max_leaf = 0;
    for (const int2 &lf : tleaves)
        if (lf.y > c->tlas_max_leaf)
            c->tlas_max_leaf = lf.y;
    return PTRT_OK;
}

void drop_graphs(ptrt_ctx *c);

void free_scene(ptrt_ctx *c) {
    drop_graphs(c); // captured launch sequences hold arena pointers
    dfree(c->d_mesh_recs);
    dfree(c->d_nodes);
    dfree(c->d_nodes2);
    dfree(c->d_tris);
    dfree(c->d_tlas_nodes);
    dfree(c->d_leaves);
    dfree(c->d_tlas_leaves);
    dfree(c->d_tlas_mesh_ids);
    dfree(c->d_tlas_heads);
    dfree(c->d_inst_pre);
    dfree(c->d_tlas_root_box);
    dfree(c->d_verts);
    dfree(c->d_slot_face);
    dfree(c->d_leaf_dst);
    dfree(c->d_node_dst);
    dfree(c->d_level_nodes);
    dfree(c->d_face_src);
    dfree(c->d_slot_pos);
    for (int k = 0; k < 2; ++k) {
        dfree(c->d_sort_keys[k]);
        dfree(c->d_sort_vals[k]);
    }
    dfree(c->d_sort_hist);
    dfree(c->d_cbounds);
    dfree(c->d_centroids);
    c->sort_capacity = 0;
}

// flags bit1 (skipped by shadow rays) comes from the materials; re-applied on either upload.
// `full` re-sends the whole records (geometry upload); otherwise only the flag words are patched
// so that boxes moved by ptrt_refit on the device are not overwritten with stale host copies.
int push_mesh_recs(ptrt_ctx *c, bool full) {
    for (int m = 0; m < c->n_meshes; ++m) {
        int flags;
        std::memcpy(&flags, &c->h_mesh_recs[(size_t)m * pt::MESH_REC_F4 + 1].w, 4);
        flags &= ~2;
        if (m < (int)c->h_shadow_skip.size() && c->h_shadow_skip[m])
            flags |= 2;
        c->h_mesh_recs[(size_t)m * pt::MESH_REC_F4 + 1].w = as_f(flags);
    }
    if (full)
        return upload(c, c->d_mesh_recs, c->h_mesh_recs);
    for (int m = 0; m < c->n_meshes; ++m)
        HIP_TRY(c, hipMemcpyAsync(&c->d_mesh_recs[(size_t)m * pt::MESH_REC_F4 + 1].w,
                                  &c->h_mesh_recs[(size_t)m * pt::MESH_REC_F4 + 1].w, 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PTRT_OK;
}

int set_device(ptrt_ctx *c) {
    HIP_TRY(c, hipSetDevice(c->device));
    return PTRT_OK;
}

int merged_pair_cap(const ptrt_ctx *c);
pt::KParams make_params(ptrt_ctx *c) {
    pt::KParams K{};
    K.mesh_recs = c->d_mesh_recs;
    K.nodes = c->d_nodes;
    K.nodes2 = c->d_nodes2;
    K.leaves = c->d_leaves;
    K.tris = c->d_tris;
    K.slot_face = c->d_slot_face;
    K.tlas_nodes = c->d_tlas_nodes;
    K.tlas_leaves = c->d_tlas_leaves;
    K.tlas_mesh_ids = c->d_tlas_mesh_ids;
    K.tlas_heads = c->d_tlas_heads;
    K.inst_c2 = c->inst_c2;
    K.materials = c->d_materials;
    K.lights = c->d_lights;
    K.blue_noise = c->d_blue;
    K.tlas_root_box = c->d_tlas_root_box;
    K.tlas_root_ref = c->tlas_root_ref;
    K.n_meshes = c->n_meshes;
    K.n_lights = c->n_lights;
    K.stack_entries = c->stack_entries;
    K.pair_meshes = c->pair_meshes;
    K.pair_tri_slots = c->pair_tri_slots;
    K.pair_max_leaf = c->pair_max_leaf;
    K.tlas_max_leaf = c->tlas_max_leaf;
    K.tlas_depth = c->tlas_depth < 1 ? 1 : c->tlas_depth;
    K.tlas_any_rounds = (c->n_meshes > 1024 || c->tlas_rounds) ? 1 : 0; // (TLAS indices beyond 10 bits do not fit a 16-bit pair entry)
    K.pair_split = (c->pair_split && !c->any_transform) ? 1 : 0;
    K.fetch_min = c->fetch_min > 0 ? c->fetch_min : 64; // 0 = refill only when the whole wave is idle: batches of 64
    K.pair_cap = merged_pair_cap(c);
    K.n_nodes = c->n_nodes;
    // the compacted leaf phase lists up to 64 x (largest leaf) tests in LEAF_PAIR_BYTES - 512 bytes of LDS: the reference
    // builder's leaves (<= 17 triangles) fit; a scene built with a larger leaf target walks its leaves lane by lane
    K.leaf_pairs = (c->leaf_pairs && (size_t)c->pair_max_leaf * 64 <= (size_t)pt::LEAF_PAIR_BYTES - 512) ? 1 : 0;
    K.leaf_min = c->leaf_min;
    K.steal = c->steal;
    K.csteal = c->csteal;
    K.csteal_min = c->csteal_min;
    K.csteal_follow = c->csteal_follow;
    K.csteal_leaf_min = c->csteal_leaf_min;
    K.cam = c->cam;
    K.sky_top = c->sky_top;
    K.sky_bottom = c->sky_bottom;
    K.use_sky = c->use_sky;
    K.env = c->d_env;
    K.env_w = c->env_w;
    K.env_h = c->env_h;
    // at a reduced render size (full-frame contexts only) the frame is rw x rh in the d_scaled_* set; the
    // generator states stay where they are: pixel p of the small frame uses state p (scene.cuh:1091-1098)
    const bool scaled = c->scaled();
    K.width = c->rw;
    K.height = c->rh;
    K.y0 = c->y0;
    K.il_period = c->il_period;
    K.il_phase = c->il_phase;
    K.rows = scaled ? c->rh : c->rows;
    K.tiles_x = (c->rw + 7) / 8;
    K.rng = c->d_rng;
    K.rng_plane = c->npix;
    K.accum = scaled ? c->s_accum : c->d_accum;
    K.normal = scaled ? c->s_normal : c->d_normal;
    K.depth = scaled ? c->s_depth : c->d_depth;
    K.object_id = scaled ? c->s_object_id : c->d_object_id;
    K.rgb8 = c->d_rgb8;
    K.counters = nullptr;
    K.tile_run = 0;
    return K;
}

int pick_geom(ptrt_ctx *c) {
    int g = c->tlas_single_leaf ? (c->all_single_leaf ? 0 : 1) : 2;
    if (c->force_geom > g)
        g = c->force_geom; // a more general variant is always valid
    return g;
}

// One 8x8 tile per 64-thread workgroup.  The frame's tile rows may be dealt to `c->split_eff` launches that run concurrently
// on the context's stream and its auxiliary streams (forked and joined by events around them: render_split_begin / _end).
template <int GEOM, int PMODE> int launch_trace(ptrt_ctx *c, const pt::KParams &K0, bool full, int grid, size_t lds) {
    const int n = c->split_eff > 1 ? c->split_eff : 1;
    const int tiles_y = grid / K0.tiles_x;
    const int slot = (int)(c->launches % EV_RING);
    const bool timed = c->time_launches && n > 1; // (one launch on the context's stream is what the frame's own events bracket)
    c->launch_timed[slot] = 0;
    for (int i = 0; i < n; ++i) {
        pt::KParams K = K0;
        K.split_n = n;
        K.split_i = i;
        const int g = n > 1 ? K0.tiles_x * ((tiles_y - i + n - 1) / n) : grid;
        if (g <= 0)
            continue;
        hipStream_t st = n > 1 ? c->aux_stream[i] : c->stream;
        hipEvent_t *ev = nullptr;
        if (timed) {
            if (c->launch_ev[i].empty()) {
                c->launch_ev[i].assign(3 * EV_RING, nullptr);
                for (auto &e : c->launch_ev[i])
                    HIP_TRY(c, hipEventCreate(&e));
            }
            ev = &c->launch_ev[i][3 * slot];
            HIP_TRY(c, hipEventRecord(ev[0], st));
            c->launch_timed[slot] = (unsigned char)(c->launch_timed[slot] | (1u << i));
        }
        bool launched = false;
        if constexpr (PMODE == 1) {
            if (c->refill_eff) {
                // Lane refill: persistent waves -- as many as the chip holds at this variant's occupancy (option "persist":
                // waves per CU) -- that draw the launch's tiles from a queue; the image is tonemapped by a pass behind them.
                K.n_tiles = g;
                K.ticket_tiles = c->ticket_tiles;
                K.queue = c->d_queue + 2 * (n > 1 ? 1 + i : 0); // (launches that share a queue are ordered: one stream each)
                if (K.counters) // (slots of its own: a launch of the one-tile kernel on another stream may still be adding to the tile slots)
                    K.counters += c->refill_counter_base * pt::COUNTER_WORDS;
                const int per_cu = c->persist > 0 ? c->persist : 4 * pt::waves_per_simd(PMODE, full, 1);
                const int waves = std::min((g + c->ticket_tiles - 1) / c->ticket_tiles, c->n_cus * per_cu);
                if (full)
                    hipLaunchKernelGGL((pt::path_trace_kernel<GEOM, true, PMODE, 1, true>), dim3(waves), dim3(64), lds, st, K);
                else
                    hipLaunchKernelGGL((pt::path_trace_kernel<GEOM, false, PMODE, 1, true>), dim3(waves), dim3(64), lds, st, K);
                if (ev)
                    HIP_TRY(c, hipEventRecord(ev[1], st));
                if (K.rgb8) {
                    hipStream_t ts = st;
                    if ((c->tm_prio & 1) && n > 1) { // the pass on a stream of the highest priority, between two events of the launch's stream
                        if (!c->tm_stream[i]) {
                            int lo = 0, hi = 0;
                            HIP_TRY(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
                            HIP_TRY(c, hipStreamCreateWithPriority(&c->tm_stream[i], hipStreamNonBlocking, hi));
                            HIP_TRY(c, hipEventCreateWithFlags(&c->tm_fork[i], hipEventDisableTiming));
                            HIP_TRY(c, hipEventCreateWithFlags(&c->tm_join[i], hipEventDisableTiming));
                        }
                        ts = c->tm_stream[i];
                        HIP_TRY(c, hipEventRecord(c->tm_fork[i], st));
                        HIP_TRY(c, hipStreamWaitEvent(ts, c->tm_fork[i], 0));
                    }
                    if (c->tm_prio & 2)
                        hipLaunchKernelGGL(pt::tonemap_tiles_kernel<true>, dim3(g), dim3(64), 0, ts, K);
                    else
                        hipLaunchKernelGGL(pt::tonemap_tiles_kernel<false>, dim3(g), dim3(64), 0, ts, K);
                    if (ts != st) {
                        HIP_TRY(c, hipEventRecord(c->tm_join[i], ts));
                        HIP_TRY(c, hipStreamWaitEvent(st, c->tm_join[i], 0));
                    }
                }
                if (ev)
                    HIP_TRY(c, hipEventRecord(ev[2], st));
                launched = true;
            }
        }
        if (launched)
            continue;
        if (full)
            hipLaunchKernelGGL((pt::path_trace_kernel<GEOM, true, PMODE>), dim3(g), dim3(64), lds, st, K);
        else
            hipLaunchKernelGGL((pt::path_trace_kernel<GEOM, false, PMODE>), dim3(g), dim3(64), lds, st, K);
        if (ev) {
            HIP_TRY(c, hipEventRecord(ev[1], st));
            HIP_TRY(c, hipEventRecord(ev[2], st));
        }
    }
    return PTRT_OK;
}

// in-wave (ray, mesh) pair compaction needs every BLAS to be one leaf and the staged
// triangle packets to fit a modest LDS budget
// PMODE 4 keeps extension and shadow pairs in one list: 64 * meshes entries always fit the extension pairs; the
// shadow pairs get what is left of a 10-KB LDS budget (16 waves per CU), at least 64 (one mesh per pass), at most
// another 64 * meshes (everything in one pass)
int merged_pair_cap(const ptrt_ctx *c) {
    const size_t rest = (size_t)c->pair_meshes * 32 + 512 + 256 + (size_t)c->stack_entries * 64 * sizeof(uint2) + pt::LEAF_PAIR_BYTES + 24;
    const int lo = 64 * c->pair_meshes + 64, hi = 128 * c->pair_meshes;
    int cap = rest < 10240 ? (int)((10240 - rest) / 2) / 64 * 64 : 0;
    cap = cap < lo ? lo : cap;
    return cap > hi ? hi : cap;
}
size_t pair_lds_bytes(const ptrt_ctx *c, int pmode) {
    if (pmode == 4)
        return (size_t)c->pair_meshes * 32 + 512 + 256 + (size_t)merged_pair_cap(c) * 2 +
               (size_t)c->stack_entries * 64 * sizeof(uint2) + pt::LEAF_PAIR_BYTES + 24;
    if (pmode == 3) // no mesh table; pair list for one TLAS leaf per ray; TLAS stack + the rays' leaf starts
        return ((size_t)c->tlas_max_leaf * 64 + pt::TLAS_FILL_TARGET) * 2 + 512 * pt::TLAS_SLOTS +
               (size_t)c->stack_entries * 64 * sizeof(uint2) + (size_t)(c->tlas_depth < 1 ? 1 : c->tlas_depth) * 512 +
               256 * pt::TLAS_SLOTS + pt::LEAF_PAIR_BYTES + 24;
    // staged heads (PMODE 1: and the mesh table), 16-bit pair entries, the rays' minima (whose second half holds the any-hit flags)
    const size_t common = (size_t)c->pair_meshes * (pmode == 1 ? 48 : 32) + (pmode == 1 ? pt::pm1_pair_bytes(c->pair_meshes) : (size_t)c->pair_meshes * 128) + 512;
    return pmode == 1 ? common + (size_t)c->pair_tri_slots * 48 + (size_t)c->pair_meshes * pt::PAIR_PAD * 16 + 16
                      : common + (size_t)c->stack_entries * 64 * sizeof(uint2) + pt::LEAF_PAIR_BYTES + 24; // (+ the ray totals)
}
// 0 lock-step, 1 pairs over single-leaf BLASes, 2 pairs over general BLASes (single-leaf TLAS)
int pair_mode(const ptrt_ctx *c, int geom, bool merged) {
    if (!c->pair_trace)
        return 0;
    if (geom == 2) // a real TLAS: rounds of one leaf per ray (pt_render.hip.h)
        return (c->tlas_max_leaf > 0 && c->tlas_max_leaf <= 32 && c->pair_tri_slots < (1 << 24) && pair_lds_bytes(c, 3) <= 40 * 1024)
                   ? 3 : 0;
    if (c->pair_meshes <= 0)
        return 0;
    if (geom == 0 && c->pair_meshes < 1024 && c->pair_max_leaf < 65536 && pair_lds_bytes(c, 1) <= 40 * 1024)
        return 1;
    // (PMODE 4's compacted leaf phase is not optional: scenes with leaves beyond its list keep PMODE 2)
    if (geom <= 1 && merged && c->leaf_pairs && c->pair_meshes < 256 && c->pair_tri_slots < (1 << 24) &&
        (size_t)c->pair_max_leaf * 64 <= (size_t)pt::LEAF_PAIR_BYTES - 512 && pair_lds_bytes(c, 4) <= 40 * 1024)
        return 4;
    if (geom <= 1 && c->pair_meshes < 256 && c->pair_tri_slots < (1 << 24) && pair_lds_bytes(c, 2) <= 40 * 1024)
        return 2;
    return 0;
}

// ---- wavefront stages ------------------------------------------------------------------------
constexpr int WF_MAX_ITERS = 16 * 17 + 2; // spp and max_depth are clamped to 16 by the Scene mirror; larger frames fall back

bool wavefront_applicable(const ptrt_ctx *c, int spp, int max_depth) {
    if (!c->wavefront || !c->tlas_single_leaf || c->pair_meshes <= 0 || c->pair_meshes > 64)
        return false;
    if (spp > 255 || max_depth > 255 || spp * (max_depth + 1) + 2 > WF_MAX_ITERS)
        return false;
    return (size_t)4 * ((size_t)c->stack_entries * 64 + pt::WF_RING / 2) * sizeof(uint2) <= 64 * 1024;
}

int run_wavefront(ptrt_ctx *c, const pt::KParams &K, bool full, int spp, int max_depth) {
    const int tiles_y = (K.rows + 7) / 8;
    const size_t items = (size_t)K.tiles_x * tiles_y * 64;
    if (items > c->wf_items) {
        dfree(c->wf_st);
        dfree(c->wf_occ);
        dfree(c->wf_planes);
        dfree(c->wf_hit);
        c->wf_items = 0;
        HIP_TRY(c, hipMalloc((void **)&c->wf_st, items * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc((void **)&c->wf_occ, items * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc((void **)&c->wf_planes, items * 25 * sizeof(float)));
        HIP_TRY(c, hipMalloc((void **)&c->wf_hit, items * sizeof(float4)));
        c->wf_items = items;
    }
    if (!c->wf_live)
        HIP_TRY(c, hipMalloc((void **)&c->wf_live, WF_MAX_ITERS * sizeof(uint32_t)));
    const size_t lds = (size_t)4 * ((size_t)c->stack_entries * 64 + pt::WF_RING / 2) * sizeof(uint2);
    if (!c->wf_trace_blocks || c->wf_trace_lds != lds) {
        int per_cu = 0, cus = 0;
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt::wf_trace_kernel, 256, lds));
        HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        if (per_cu < 1 || cus < 1)
            return fail(c, PTRT_E_HIP, "wavefront trace kernel does not fit (LDS %zu bytes)", lds);
        c->wf_trace_blocks = per_cu * cus;
        c->wf_trace_lds = lds;
    }
    pt::WfParams W{};
    W.st = c->wf_st;
    W.occ = c->wf_occ;
    W.live = c->wf_live;
    W.hit = c->wf_hit;
    float *p = c->wf_planes;
    W.ray = p;
    W.thr = p + 6 * items;
    W.acc = p + 9 * items;
    W.avg = p + 12 * items;
    W.pend = p + 15 * items;
    W.sh = p + 18 * items;
    W.n_items = (int)items;
    W.fetch_min = c->fetch_min > 0 ? c->fetch_min : 16;
    const int iters = spp * (max_depth + 1);
    HIP_TRY(c, hipMemsetAsync(c->wf_live, 0, (size_t)(iters + 2) * sizeof(uint32_t), c->stream));
    const int shade_blocks = (int)((items + 255) / 256);
    const int chunks = (int)(items / 64);
    const int trace_blocks = std::min(c->wf_trace_blocks, (chunks + 3) / 4);
    for (int it = 0; it <= iters; ++it) {
        W.iter = it;
        if (it > 0)
            hipLaunchKernelGGL(pt::wf_trace_kernel, dim3(trace_blocks), dim3(256), lds, c->stream, K, W);
        if (c->wf_sort && it > 0) { // (the first shade only regenerates: one class)
            const int G = c->wf_sort;
            const int sb = (int)((items + 256 * (size_t)G - 1) / (256 * (size_t)G));
            auto kern = full ? (G == 1 ? pt::wf_shade_kernel<true, 1> : G == 2 ? pt::wf_shade_kernel<true, 2> : pt::wf_shade_kernel<true, 4>)
                             : (G == 1 ? pt::wf_shade_kernel<false, 1> : G == 2 ? pt::wf_shade_kernel<false, 2> : pt::wf_shade_kernel<false, 4>);
            hipLaunchKernelGGL(kern, dim3(sb), dim3(256), 0, c->stream, K, W);
        } else if (full)
            hipLaunchKernelGGL(pt::wf_shade_kernel<true>, dim3(shade_blocks), dim3(256), 0, c->stream, K, W);
        else
            hipLaunchKernelGGL(pt::wf_shade_kernel<false>, dim3(shade_blocks), dim3(256), 0, c->stream, K, W);
    }
    HIP_TRY(c, hipGetLastError());
    return PTRT_OK;
}

// ---- asynchronous-lane megakernel -------------------------------------------------------------
bool async_applicable(const ptrt_ctx *c) {
    if (!c->async_lanes || !c->tlas_single_leaf || c->pair_meshes <= 0 || c->pair_meshes > 64)
        return false;
    if ((size_t)c->pair_max_leaf * 64 > (size_t)pt::LEAF_PAIR_BYTES - 512) // its leaf phase is always the compacted one
        return false;
    return ((size_t)c->stack_entries * 64 + pt::AS_RING / 2) * sizeof(uint2) + pt::LEAF_PAIR_BYTES <= 40 * 1024;
}

int run_async(ptrt_ctx *c, const pt::KParams &K, bool full) {
    const size_t lds = ((size_t)c->stack_entries * 64 + pt::AS_RING / 2) * sizeof(uint2) + pt::LEAF_PAIR_BYTES;
    if (!c->as_cursor)
        HIP_TRY(c, hipMalloc((void **)&c->as_cursor, sizeof(uint32_t)));
    if (c->as_lds != lds || !c->as_blocks[full ? 1 : 0]) {
        if (c->as_lds != lds)
            c->as_blocks[0] = c->as_blocks[1] = 0;
        int per_cu = 0, cus = 0;
        if (full)
            HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt::path_trace_async_kernel<true>, 64, lds));
        else
            HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt::path_trace_async_kernel<false>, 64, lds));
        HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        if (per_cu < 1 || cus < 1)
            return fail(c, PTRT_E_HIP, "asynchronous trace kernel does not fit (LDS %zu bytes)", lds);
        c->as_blocks[full ? 1 : 0] = per_cu * cus;
        c->as_lds = lds;
    }
    pt::AsyncParams A{};
    A.cursor = c->as_cursor;
    A.n_tiles = K.tiles_x * ((K.rows + 7) / 8);
    A.shade_min = c->shade_min;
    A.leaf_min = c->as_leaf_min;
    const int grid = std::min(c->as_blocks[full ? 1 : 0], A.n_tiles);
    HIP_TRY(c, hipMemsetAsync(c->as_cursor, 0, sizeof(uint32_t), c->stream));
    if (full)
        hipLaunchKernelGGL(pt::path_trace_async_kernel<true>, dim3(grid), dim3(64), lds, c->stream, K, A);
    else
        hipLaunchKernelGGL(pt::path_trace_async_kernel<false>, dim3(grid), dim3(64), lds, c->stream, K, A);
    HIP_TRY(c, hipGetLastError());
    return PTRT_OK;
}

// `device_work`: the caller may enqueue work on the context's stream or change what its kernels read from memory -- every
// entry point except the few host-only ones below.  The next ptrt_render then orders its launches behind the stream (see
// "pipeline" there) instead of overlapping them with the previous frame's.
bool ctx_live(ptrt_ctx *c, bool device_work = true) {
    std::lock_guard<std::mutex> lock(g_live_mutex);
    const bool live = c && g_live.count(c);
    if (live && device_work) {
        c->touched = true;
        c->heads_fresh = false;
    }
    return live;
}

// ---- dynamic geometry: launch sequences, replayed as hipGraphs ------------------------------
// A refit is ~16 and a rebuild ~35 microsecond-sized launches; issued one by one they cost more in
// launch gaps than in work.  The sequences are fixed for a given upload (same kernels, grids and
// arena pointers), so each is captured once on a private stream and replayed with one
// hipGraphLaunch on the context's stream.  Key -1 = refit, key m >= 0 = rebuild of mesh m + refit.
void drop_graphs(ptrt_ctx *c) {
    for (auto &kv : c->graphs)
        (void)hipGraphExecDestroy(kv.second);
    c->graphs.clear();
}

// the two-level records follow the canonical nodes (after an upload, a refit, a rebuild)
void enqueue_expand_nodes(ptrt_ctx *c, hipStream_t st) {
    if (PT_TWO_LEVEL && c->n_nodes > 0 && c->d_nodes2)
        hipLaunchKernelGGL(pt::expand_nodes_kernel, dim3((c->n_nodes * 3 + 255) / 256), dim3(256), 0, st, c->d_nodes, c->d_nodes2,
                           c->n_nodes);
}

int enqueue_refit(ptrt_ctx *c, hipStream_t st) {
    const int B = 256;
    if (c->n_slots > 0)
        hipLaunchKernelGGL(pt::repack_tris_kernel, dim3((c->n_slots + B - 1) / B), dim3(B), 0, st, c->d_verts,
                           c->d_slot_face, c->d_tris, c->n_slots);
    if (c->n_leaves > 0)
        hipLaunchKernelGGL(pt::refit_leaves_kernel, dim3((c->n_leaves + B - 1) / B), dim3(B), 0, st, c->d_verts,
                           c->d_slot_face, c->d_leaves, c->d_leaf_dst, c->d_nodes, c->d_mesh_recs, c->n_leaves);
    // wide levels: one launch each, deepest first; the narrow levels near the root (<= 2048 nodes) and the
    // TLAS root box: one workgroup, barriers between levels (a tiny launch costs ~4.6 us whatever it does)
    pt::TopLevels T{};
    bool top = false;
    for (int d = (int)c->level_offset.size() - 1; d >= 1; --d) { // depth d nodes: [offset[d-1], offset[d])
        const int begin = c->level_offset[d - 1], count = c->level_offset[d] - begin;
        if (count <= 0)
            continue;
        top = top || count <= 2048; // counts shrink towards the root; once narrow, the rest goes to the fused kernel
        if (top) {
            T.begin[T.n] = begin;
            T.count[T.n] = count;
            ++T.n;
        } else {
            hipLaunchKernelGGL(pt::refit_level_kernel, dim3((count + B - 1) / B), dim3(B), 0, st, c->d_level_nodes + begin,
                               count, c->d_node_dst, c->d_nodes, c->d_mesh_recs);
        }
    }
    hipLaunchKernelGGL(pt::refit_top_levels_kernel, dim3(1), dim3(1024), 0, st, c->d_level_nodes, T, c->d_node_dst,
                       c->d_nodes, c->d_mesh_recs, c->d_tlas_leaves, c->d_tlas_mesh_ids, c->tlas_root_ref,
                       c->d_tlas_root_box);
    enqueue_expand_nodes(c, st);
    HIP_TRY(c, hipGetLastError());
    return PTRT_OK;
}

int enqueue_build(ptrt_ctx *c, int mesh, hipStream_t st) {
    const int n = c->mesh_face_count[mesh];
    const int n_waves = (n + pt::RS_WAVE_KEYS - 1) / pt::RS_WAVE_KEYS;
    const int B = 256, G = (n + B - 1) / B;
    const int4 *faces = c->d_face_src + c->mesh_face_base[mesh];
    hipLaunchKernelGGL(pt::centroid_bounds_kernel, dim3(G < 128 ? G : 128), dim3(B), 0, st, c->d_verts, faces, n,
                       c->d_centroids, c->d_cbounds);
    hipLaunchKernelGGL(pt::morton_kernel, dim3(G), dim3(B), 0, st, c->d_centroids, c->d_cbounds, n, c->d_sort_keys[0],
                       c->d_sort_vals[0]);
    const int wg = (n_waves + pt::RS_BLOCK / 64 - 1) / (pt::RS_BLOCK / 64);
    int cur = 0;
    for (int shift = 0; shift < 32; shift += 8) { // the top pass only sees bits 24..29 of the 30-bit code
        hipLaunchKernelGGL(pt::rs_hist_kernel, dim3(wg), dim3(pt::RS_BLOCK), 0, st, c->d_sort_keys[cur], n, shift,
                           c->d_sort_hist, n_waves);
        uint32_t *positions = c->d_sort_hist + (size_t)n_waves * 256;
        hipLaunchKernelGGL(pt::rs_scan_kernel, dim3(1), dim3(1024), 0, st, c->d_sort_hist, positions, n_waves);
        hipLaunchKernelGGL(pt::rs_scatter_kernel, dim3(wg), dim3(pt::RS_BLOCK), 0, st, c->d_sort_keys[cur],
                           c->d_sort_vals[cur], n, shift, positions, n_waves, c->d_sort_keys[cur ^ 1],
                           c->d_sort_vals[cur ^ 1]);
        cur ^= 1;
    }
    hipLaunchKernelGGL(pt::apply_order_kernel, dim3(G), dim3(B), 0, st, c->d_sort_vals[cur], c->d_slot_pos, faces,
                       c->d_slot_face, c->mesh_slot_base[mesh], n, c->d_cbounds);
    HIP_TRY(c, hipGetLastError());
    return PTRT_OK;
}

template <class F> int run_graphed(ptrt_ctx *c, int key, F enqueue) {
    if (!c->use_graphs)
        return enqueue(c->stream);
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
        if (!c->capture_stream)
            HIP_TRY(c, hipStreamCreateWithFlags(&c->capture_stream, hipStreamNonBlocking));
        HIP_TRY(c, hipStreamBeginCapture(c->capture_stream, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue(c->capture_stream);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(c->capture_stream, &g);
        if (rc != PTRT_OK || e != hipSuccess || !g) {
            if (g)
                (void)hipGraphDestroy(g);
            return rc != PTRT_OK ? rc : fail(c, PTRT_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        }
        hipGraphExec_t exec = nullptr;
        const hipError_t ei = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ei != hipSuccess)
            return fail(c, PTRT_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ei));
        it = c->graphs.emplace(key, exec).first;
    }
    HIP_TRY(c, hipGraphLaunch(it->second, c->stream));
    return PTRT_OK;
}

void free_denoiser(ptrt_ctx *c) {
    dfree(c->dn_cur4);
    for (int k = 0; k < 2; ++k) {
        dfree(c->dn_c4[k]);
        dfree(c->dn_g4[k]);
        dfree(c->dn_h1[k]);
        dfree(c->dn_h2[k]);
    }
    dfree(c->dn_hobj);
    dfree(c->dn_motion);
    dfree(c->dn_out);
    dfree(c->dn_pvp);
    c->dn_on = false;
}

// motion vectors -> Denoiser::denoise (non-split) -> tonemap of the denoised image, all on the
// context's stream (Scene::render_to_device, scene.cuh:1103-1127,1204).  History hand-over is a
// swap of the double-buffered sets (history moments AND the packed G-buffer); no device copies.
// 3 + atrous_iterations launches per frame: prep, temporal, variance, a-trous x N (the last one
// also writes the API's vec3 image and the RGB8 frame).
int run_denoiser(ptrt_ctx *c, const pt::KParams &K, unsigned char *rgb8) {
    const int W = c->rw, H = c->rh; // the render size (the denoiser was allocated for it)
    const dim3 grid((W + 63) / 64, (H + 3) / 4), block(256);
    const pt::DenoiseSettings &S = c->dn;
    const int prev = c->dn_cur, next = c->dn_cur ^ 1;
    // perfSettings.enableMotionVectors (scene.cuh:1103); when off the last vectors are reused
    if (c->mv_active)
        HIP_TRY(c, hipMemcpyAsync(c->dn_pvp, c->prev_view_proj, 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pt::prep_kernel, grid, block, 0, c->stream, c->dn_g4[next], c->dn_motion, c->dn_cur4, K.accum,
                       K.normal, K.depth, W, H, c->cam.origin, c->cam.llc, c->cam.horizontal, c->cam.vertical, c->cam.u,
                       c->cam.v, c->cam.lens_radius, c->dn_pvp, c->mv_active, S.sky_depth_threshold, S.enable_firefly_suppression);
    hipLaunchKernelGGL(pt::temporal_kernel, grid, block, 0, c->stream, c->dn_h1[next], c->dn_h2[next], c->dn_cur4,
                       c->dn_h1[prev], c->dn_h2[prev], c->dn_motion, c->dn_g4[next], c->dn_g4[prev], K.object_id,
                       c->dn_hobj, S, c->dn_first ? 1 : 0, W, H);
    c->dn_cur = next;
    hipLaunchKernelGGL(pt::variance_kernel, grid, block, 0, c->stream, c->dn_c4[0], c->dn_h1[next], c->dn_h2[next],
                       c->dn_g4[next], K.object_id, c->dn_hobj, S.sky_depth_threshold, S.use_object_ids, W, H);
    const int steps[5] = {1, 2, 4, 8, 16};
    const int iters = S.atrous_iterations < 5 ? (S.atrous_iterations < 0 ? 0 : S.atrous_iterations) : 5;
    for (int i = 0; i < iters; ++i) {
        const float4 *in = c->dn_c4[i & 1];
        float4 *out = c->dn_c4[(i + 1) & 1];
        // a workgroup = 64 consecutive pixels x 4 rows of ONE row class (y mod step); the class is the fast block index
        const int s = steps[i];
        const dim3 agrid((W + pt::AT_W - 1) / pt::AT_W, (((H + s - 1) / s + pt::AT_ROWS - 1) / pt::AT_ROWS) * s);
        const size_t alds = pt::atrous_lds_bytes(s);
        const bool last = i == iters - 1;
        auto kern = last ? (c->atrous_exp ? pt::atrous_kernel<true, true> : pt::atrous_kernel<true, false>)
                         : (c->atrous_exp ? pt::atrous_kernel<false, true> : pt::atrous_kernel<false, false>);
        hipLaunchKernelGGL(kern, agrid, block, alds, c->stream, out, in, c->dn_g4[next], K.object_id, steps[i], S.sigma_luminance,
                           S.sky_depth_threshold, S.edge_depth_threshold, S.edge_normal_threshold, S.use_object_ids, W, H,
                           last ? c->dn_out : (float *)nullptr, last ? rgb8 : (unsigned char *)nullptr);
    }
    if (iters == 0)
        hipLaunchKernelGGL(pt::c4_to_output_kernel, grid, block, 0, c->stream, c->dn_c4[0], W, H, c->dn_out, rgb8);
    HIP_TRY(c, hipGetLastError());
    c->dn_first = false;
    return PTRT_OK;
}

// Step 5 of Scene::render_to_device (scene.cuh:1137-1183) on `image` (w x h, in place), with the
// reference's own pass list and sizes (see pt_post.hip.h); `rgb8` non-NULL: also tonemap the result.
int run_bloom(ptrt_ctx *c, float *image, int w, int h, unsigned char *rgb8) {
    std::vector<pt::BloomPass> passes;
    int mip_w = w, mip_h = h;
    const float *last = image;
    for (int i = 0; i < 6; ++i) {
        const int next_w = mip_w / 2, next_h = mip_h / 2;
        passes.push_back(pt::BloomPass{i == 0 ? 0 : 1, c->bl_mip[i], last, mip_w, mip_h, next_w, next_h});
        last = c->bl_mip[i];
        mip_w = next_w;
        mip_h = next_h;
    }
    for (int i = 4; i >= 0; --i) {
        mip_w *= 2;
        mip_h *= 2;
        passes.push_back(pt::BloomPass{2, c->bl_mip[i], c->bl_mip[i + 1], mip_w / 2, mip_h / 2, (mip_w / 2) * 2, (mip_h / 2) * 2});
    }
    const bool fuse_final = (w % 2) == 0; // else the reference's row stride 2*(w/2) is not the frame's
    if (!fuse_final)
        passes.push_back(pt::BloomPass{2, image, c->bl_mip[0], w / 2, h / 2, (w / 2) * 2, (h / 2) * 2});
    for (size_t k = 0; k < passes.size();) {
        // one workgroup beats a launch (~6 us) only for the smallest levels: measured 46 us for five passes of
        // up to 8 K pixels vs ~30 us as separate launches
        auto small = [&](size_t i) { return (size_t)passes[i].out_w * passes[i].out_h <= 4096; };
        if (small(k)) {
            pt::BloomSmallPasses S{};
            while (k < passes.size() && small(k) && S.n < 8)
                S.p[S.n++] = passes[k++];
            hipLaunchKernelGGL(pt::bloom_small_passes_kernel, dim3(1), dim3(1024), 0, c->stream, S);
        } else {
            const pt::BloomPass &P = passes[k++];
            hipLaunchKernelGGL(pt::bloom_pass_kernel, dim3((P.out_w + 63) / 64, (P.out_h + 3) / 4), dim3(256), 0, c->stream, P);
        }
    }
    const dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
    if (fuse_final)
        hipLaunchKernelGGL(pt::bloom_final_kernel, grid, block, 0, c->stream, image, c->bl_mip[0], w, h, w / 2, h / 2, rgb8);
    else if (rgb8)
        hipLaunchKernelGGL(pt::tonemap_only_kernel, grid, block, 0, c->stream, rgb8, image, w, h);
    HIP_TRY(c, hipGetLastError());
    return PTRT_OK;
}

void free_present(ptrt_ctx *c) {
    for (auto &s : c->present) {
        if (s.dev)
            (void)hipFree(s.dev);
        if (s.host)
            (void)hipHostFree(s.host);
        if (s.rendered)
            (void)hipEventDestroy(s.rendered);
        if (s.done)
            (void)hipEventDestroy(s.done);
    }
    c->present.clear();
    if (c->present_stream) {
        (void)hipStreamSynchronize(c->present_stream);
        (void)hipStreamDestroy(c->present_stream);
        c->present_stream = nullptr;
    }
}

void free_post(ptrt_ctx *c) {
    dfree(c->s_accum);
    dfree(c->s_normal);
    dfree(c->s_depth);
    dfree(c->s_object_id);
    for (int i = 0; i < 6; ++i)
        dfree(c->bl_mip[i]);
    c->bloom_on = 0;
}

} // namespace

// =====================================================================================
extern "C" {

int ptrt_abi_version(void) { return PTRT_ABI_VERSION; }

const char *ptrt_last_error(const ptrt_ctx *ctx) {
    if (ctx && ctx_live(const_cast<ptrt_ctx *>(ctx), false))
        return ctx->err.c_str();
    return g_last_error.c_str();
}

namespace {
int create_ctx(int full_w, int full_h, int tile_y0, int tile_rows, int il_period, int il_phase, int device, ptrt_ctx **out);
}

int ptrt_create(int full_w, int full_h, int tile_y0, int tile_rows, int device, ptrt_ctx **out) {
    if (!out)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_create: out is NULL");
    *out = nullptr;
    if (full_w <= 0 || full_h <= 0)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_create: bad frame size %dx%d", full_w, full_h);
    if (tile_rows <= 0) {
        tile_y0 = 0;
        tile_rows = full_h;
    }
    if (tile_y0 < 0 || tile_y0 + tile_rows > full_h)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_create: tile rows [%d,%d) outside 0..%d", tile_y0,
                    tile_y0 + tile_rows, full_h);
    return create_ctx(full_w, full_h, tile_y0, tile_rows, 1, 0, device, out);
}

int ptrt_create_interleaved(int full_w, int full_h, int phase, int period, int device, ptrt_ctx **out) {
    if (!out)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_create_interleaved: out is NULL");
    *out = nullptr;
    if (full_w <= 0 || full_h <= 0)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_create_interleaved: bad frame size %dx%d", full_w, full_h);
    const int strips = (full_h + 7) / 8;
    if (period < 1 || phase < 0 || phase >= period || phase >= strips)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_create_interleaved: strip %d of every %d (the frame has %d strips of 8 rows)",
                    phase, period, strips);
    if (period == 1)
        return create_ctx(full_w, full_h, 0, full_h, 1, 0, device, out);
    // rows of the strips phase, phase + period, ...; only the frame's last strip can be short, and it is the owner's last
    int rows = 0;
    for (int t = phase; t < strips; t += period)
        rows += (t * 8 + 8 <= full_h) ? 8 : full_h - t * 8;
    return create_ctx(full_w, full_h, phase * 8, rows, period, phase, device, out);
}

namespace {
int create_ctx(int full_w, int full_h, int tile_y0, int tile_rows, int il_period, int il_phase, int device, ptrt_ctx **out) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, PTRT_E_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev)
        return fail(nullptr, PTRT_E_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    ptrt_ctx *c = new ptrt_ctx();
    c->device = device;
    c->W = full_w;
    c->H = full_h;
    c->y0 = tile_y0;
    c->rows = tile_rows;
    c->il_period = il_period;
    c->il_phase = il_phase;
    c->npix = (size_t)full_w * tile_rows;
    c->rw = full_w;
    c->rh = full_h;
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        g_live.insert(c);
    }
    *out = c;
    int rc = set_device(c);
    if (rc)
        return rc;
    HIP_TRY(c, hipStreamCreate(&c->own_stream));
    c->stream = c->own_stream;
    c->ev_ring.assign(2 * EV_RING, nullptr);
    for (auto &ev : c->ev_ring)
        HIP_TRY(c, hipEventCreate(&ev));
    HIP_TRY(c, hipMalloc((void **)&c->d_rng, c->npix * 6 * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->d_accum, c->npix * 3 * sizeof(float)));
    HIP_TRY(c, hipMalloc((void **)&c->d_normal, c->npix * 3 * sizeof(float)));
    HIP_TRY(c, hipMalloc((void **)&c->d_depth, c->npix * sizeof(float)));
    HIP_TRY(c, hipMalloc((void **)&c->d_object_id, c->npix * sizeof(int)));
    HIP_TRY(c, hipMalloc((void **)&c->d_rgb8, c->npix * 3));
    c->n_counter_slots = (size_t)((c->W + 7) / 8) * ((c->rows + 7) / 8); // one slot of pt::COUNTER_WORDS per 8x8-pixel workgroup
    c->n_counter_slots += 4; // the shade stage uses one slot per wave of a 256-thread grid (rounded up)
    c->n_counter_slots += (size_t)ptrt_ctx::MAX_SPLIT * ((c->W + 7) / 8);
    // (lane refill: a slot per persistent wave and launch of a split frame -- never more waves than tiles -- in a range of their own:
    // consecutive overlapping frames may run the one-tile kernel and the refill kernel on different streams at the same time)
    c->refill_counter_base = c->n_counter_slots;
    c->n_counter_slots += (size_t)((c->W + 7) / 8) * ((c->rows + 7) / 8) + (size_t)ptrt_ctx::MAX_SPLIT * ((c->W + 7) / 8);
    HIP_TRY(c, hipMalloc((void **)&c->d_queue, 2 * (1 + ptrt_ctx::MAX_SPLIT) * sizeof(unsigned int)));
    HIP_TRY(c, hipMemsetAsync(c->d_queue, 0, 2 * (1 + ptrt_ctx::MAX_SPLIT) * sizeof(unsigned int), c->stream));
    HIP_TRY(c, hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, c->device));
    HIP_TRY(c, hipMalloc((void **)&c->d_counters, c->n_counter_slots * pt::COUNTER_WORDS * sizeof(unsigned long long)));
    HIP_TRY(c, hipMalloc((void **)&c->d_blue, PTRT_BLUE_NOISE_FLOATS * sizeof(float)));
    HIP_TRY(c, hipMemsetAsync(c->d_rng, 0, c->npix * 6 * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_accum, 0, c->npix * 3 * sizeof(float), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_rgb8, 0, c->npix * 3, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, c->n_counter_slots * pt::COUNTER_WORDS * sizeof(unsigned long long), c->stream));
    // the blue-noise table is all zeros until the application installs one (bluenoise.cuh:46,189)
    HIP_TRY(c, hipMemsetAsync(c->d_blue, 0, PTRT_BLUE_NOISE_FLOATS * sizeof(float), c->stream));
    c->last_rgb8 = c->d_rgb8;
    // default camera: Camera(aspect, 2, 1) of the Scene constructor (scene.cuh:748, camera.cuh:127-148)
    const float aspect = (float)full_w / (float)full_h;
    const float vw = 2.0f * aspect;
    c->cam.origin = pt::f3{0, 0, 0};
    c->cam.horizontal = pt::f3{vw, 0, 0};
    c->cam.vertical = pt::f3{0, 2.0f, 0};
    c->cam.llc = pt::f3{0.0f - vw * 0.5f, 0.0f - 2.0f * 0.5f, -1.0f};
    c->cam.u = pt::f3{1, 0, 0};
    c->cam.v = pt::f3{0, 1, 0};
    c->cam.w = pt::f3{0, 0, 1};
    c->cam.lens_radius = 0.0f;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PTRT_OK;
}
} // namespace

void ptrt_destroy(ptrt_ctx *c) {
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        if (!c || !g_live.count(c))
            return; // NULL or already destroyed: ignored, like a repeated cudaFree in the reference
        g_live.erase(c);
    }
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < ptrt_ctx::MAX_SPLIT; ++i) { // the auxiliary streams of split launches
        if (c->aux_stream[i]) {
            (void)hipStreamSynchronize(c->aux_stream[i]);
            (void)hipStreamDestroy(c->aux_stream[i]);
        }
        if (c->split_join[i])
            (void)hipEventDestroy(c->split_join[i]);
    }
    for (int i = 0; i < ptrt_ctx::MAX_SPLIT; ++i) {
        if (c->tm_stream[i]) {
            (void)hipStreamSynchronize(c->tm_stream[i]);
            (void)hipStreamDestroy(c->tm_stream[i]);
        }
        if (c->tm_fork[i])
            (void)hipEventDestroy(c->tm_fork[i]);
        if (c->tm_join[i])
            (void)hipEventDestroy(c->tm_join[i]);
        for (hipEvent_t e : c->launch_ev[i])
            if (e)
                (void)hipEventDestroy(e);
    }
    if (c->split_fork)
        (void)hipEventDestroy(c->split_fork);
    for (hipEvent_t e : c->head_ev)
        if (e)
            (void)hipEventDestroy(e);
    free_scene(c);
    dfree(c->d_materials);
    dfree(c->d_lights);
    dfree(c->d_rng);
    dfree(c->d_accum);
    dfree(c->d_normal);
    dfree(c->d_depth);
    dfree(c->d_object_id);
    dfree(c->alt_accum);
    dfree(c->alt_normal);
    dfree(c->alt_depth);
    dfree(c->alt_object_id);
    dfree(c->d_rgb8);
    dfree(c->d_counters);
    dfree(c->d_queue);
    for (int k = 0; k < ptrt_ctx::STAGES; ++k) {
        if (c->h_stage[k])
            (void)hipHostFree(c->h_stage[k]);
        if (c->stage_ev[k])
            (void)hipEventDestroy(c->stage_ev[k]);
        dfree(c->d_stage[k]);
    }
    if (c->copy_stream)
        (void)hipStreamDestroy(c->copy_stream);
    if (c->copy_ev)
        (void)hipEventDestroy(c->copy_ev);
    dfree(c->wf_st);
    dfree(c->wf_occ);
    dfree(c->wf_live);
    dfree(c->wf_planes);
    dfree(c->wf_hit);
    dfree(c->as_cursor);
    dfree(c->d_blue);
    dfree(c->d_jump);
    dfree(c->d_env);
    free_denoiser(c);
    free_post(c);
    free_present(c);
    for (auto &ev : c->ev_ring)
        if (ev)
            (void)hipEventDestroy(ev);
    if (c->capture_stream)
        (void)hipStreamDestroy(c->capture_stream);
    if (c->own_stream)
        (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int ptrt_set_blue_noise(ptrt_ctx *c, const float *table) {
    if (!ctx_live(c) || !table)
        return fail(c, PTRT_E_INVALID, "ptrt_set_blue_noise: bad argument");
    if (int rc = set_device(c))
        return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_blue, table, PTRT_BLUE_NOISE_FLOATS * sizeof(float), hipMemcpyHostToDevice,
                              c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PTRT_OK;
}

int ptrt_reset_rng(ptrt_ctx *c, unsigned long long seed) {
    if (!ctx_live(c))
        return fail(c, PTRT_E_INVALID, "ptrt_reset_rng: bad context");
    if (int rc = set_device(c))
        return rc;
    // bits needed for the largest global pixel index of this tile
    const unsigned long long last = (unsigned long long)(c->il_period > 1 ? c->H : c->y0 + c->rows) * (unsigned long long)c->W;
    int bits = 1;
    while ((last >> bits) != 0)
        ++bits;
    if (bits > c->n_jump) {
        if (bits > JUMP_MAX)
            return fail(c, PTRT_E_INVALID, "ptrt_reset_rng: frame of 2^%d pixels (at most 2^%d)", bits, JUMP_MAX);
        const std::vector<uint32_t> &J = jump_matrices();
        dfree(c->d_jump);
        HIP_TRY(c, hipMalloc((void **)&c->d_jump, (size_t)bits * 800 * sizeof(uint32_t)));
        HIP_TRY(c, hipMemcpy(c->d_jump, J.data(), (size_t)bits * 800 * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->n_jump = bits;
    }
    // curand_init's state scrambling (published cuRAND XORWOW; constants unverified)
    const uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    const uint32_t d0 = 6615241u + t1 + t0;
    const uint32_t v0 = 123456789u + t0, v1 = 362436069u ^ t0, v2 = 521288629u + t1, v3 = 88675123u ^ t1,
                   v4 = 5783321u + t0;
    const int grid = (int)((c->npix + 255) / 256);
    hipLaunchKernelGGL(pt::xorwow_init_kernel, dim3(grid), dim3(256), 0, c->stream, c->d_rng, c->W, c->rows, c->y0,
                       c->il_period, c->il_phase, d0, v0, v1, v2, v3, v4, c->d_jump, c->n_jump);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // the reference synchronises here too (scene.cuh:455)
    c->rng_ready = true;
    return PTRT_OK;
}

int upload_instance_pretests(ptrt_ctx *c, const ptrt_mesh_desc *meshes, int mesh_count, const float4 *device_recs = nullptr);

int ptrt_upload_geometry(ptrt_ctx *c, const ptrt_mesh_desc *meshes, int mesh_count, const ptrt_bvh_node *tlas_nodes,
                         int tlas_node_count, const int32_t *tlas_mesh_indices, int tlas_index_count) {
    if (!ctx_live(c))
        return fail(c, PTRT_E_INVALID, "ptrt_upload_geometry: bad context");
    if (!meshes || mesh_count <= 0 || !tlas_nodes || tlas_node_count <= 0 || !tlas_mesh_indices ||
        tlas_index_count <= 0)
        return fail(c, PTRT_E_INVALID, "ptrt_upload_geometry: empty scene");
    if (int rc = set_device(c))
        return rc;

    Relayout R;
    std::vector<float4> recs((size_t)mesh_count * pt::MESH_REC_F4, f4(0, 0, 0, 0));
    bool all_leaf = true;
    std::string why;
    // refit bookkeeping
    std::vector<float> all_verts;
    std::vector<int4> slot_face;
    std::vector<int> leaf_dst, node_dst, node_depth, vert_base(mesh_count), vert_count(mesh_count);
    // rebuild bookkeeping
    std::vector<int4> face_src;
    std::vector<int> slot_pos, face_base(mesh_count), face_count(mesh_count), slot_base(mesh_count), slot_count(mesh_count);
    std::vector<unsigned char> rebuildable(mesh_count, 0), is_soup(mesh_count, 0);
    for (int m = 0; m < mesh_count; ++m) {
        const ptrt_mesh_desc &M = meshes[m];
        if (!M.verts || !M.faces || !M.nodes || !M.prim_indices || M.node_count <= 0 || M.face_count <= 0 ||
            M.vert_count <= 0 || M.prim_count <= 0)
            return fail(c, PTRT_E_INVALID, "mesh %d: missing arrays (a mesh needs vertices, faces and a built BVH)", m);
        for (int f = 0; f < M.face_count; ++f) {
            const ptrt_tri &t = M.faces[f];
            if (t.v0 < 0 || t.v1 < 0 || t.v2 < 0 || t.v0 >= M.vert_count || t.v1 >= M.vert_count ||
                t.v2 >= M.vert_count)
                return fail(c, PTRT_E_INVALID, "mesh %d: face %d references a vertex out of range", m, f);
        }
        vert_base[m] = (int)(all_verts.size() / 3);
        vert_count[m] = M.vert_count;
        all_verts.insert(all_verts.end(), &M.verts[0].x, &M.verts[0].x + (size_t)M.vert_count * 3);
        const int vb = vert_base[m];
        face_base[m] = (int)face_src.size();
        face_count[m] = M.face_count;
        slot_base[m] = (int)slot_face.size();
        bool soup = M.vert_count == 3 * M.face_count;
        for (int f = 0; f < M.face_count; ++f) {
            const ptrt_tri &t = M.faces[f];
            face_src.push_back(make_int4(vb + t.v0, vb + t.v1, vb + t.v2, f));
            soup = soup && t.v0 == 3 * f && t.v1 == 3 * f + 1 && t.v2 == 3 * f + 2;
        }
        is_soup[m] = soup ? 1 : 0;
        bool bad = false;
        auto emit_leaf = [&](int start, int count, int dst) -> int {
            const int id = (int)R.leaves.size();
            if (count > 0 && (start < 0 || start + count > M.prim_count)) {
                bad = true;
                count = 0;
            }
            R.leaves.push_back(make_int2((int)(R.tris.size() / 3), count));
            leaf_dst.push_back(dst);
            for (int i = 0; i < count; ++i) {
                const int fidx = M.prim_indices[start + i];
                slot_pos.push_back(start + i);
                if (fidx < 0 || fidx >= M.face_count) {
                    bad = true;
                    R.tris.insert(R.tris.end(), 3, f4(0, 0, 0, 0));
                    slot_face.push_back(make_int4(vb, vb, vb, 0));
                    continue;
                }
                const ptrt_tri &t = M.faces[fidx];
                slot_face.push_back(make_int4(vb + t.v0, vb + t.v1, vb + t.v2, fidx));
                const ptrt_vec3 &a = M.verts[t.v0], &b = M.verts[t.v1], &d = M.verts[t.v2];
                // e1 = v1 - v0, e2 = v2 - v0: the same fp32 subtractions the reference performs per
                // test (intersection.cuh:224-225), done once here
                R.tris.push_back(f4(a.x, a.y, a.z, 0.0f)); // (the w words: tri_normals_kernel, below)
                R.tris.push_back(f4(b.x - a.x, b.y - a.y, b.z - a.z, 0.0f));
                R.tris.push_back(f4(d.x - a.x, d.y - a.y, d.z - a.z, 0.0f));
            }
            return id;
        };
        int depth = 0;
        const int root = convert_tree(M.nodes, M.node_count, R.nodes, emit_leaf, depth, why, -(m + 1), &node_dst, &node_depth);
        if (root == INT32_MIN || bad)
            return fail(c, PTRT_E_INVALID, "mesh %d: malformed BVH (%s)", m, bad ? "leaf range out of bounds" : why.c_str());
        if (depth > 23)
            return fail(c, PTRT_E_INVALID,
                        "mesh %d: BVH is %d levels deep; the traversal stack (24 entries, as in the reference) needs <= 23",
                        m, depth);
        if (depth > R.max_depth)
            R.max_depth = depth;
        if (root >= 0)
            all_leaf = false;
        // a GPU rebuild permutes faces over prim positions: every position 0..face_count-1 must be a leaf slot once
        slot_count[m] = (int)slot_face.size() - slot_base[m];
        if (slot_count[m] == M.face_count && M.prim_count == M.face_count) {
            std::vector<unsigned char> used((size_t)M.face_count, 0);
            bool once = true;
            for (int s = slot_base[m]; s < slot_base[m] + slot_count[m]; ++s) {
                once = once && !used[(size_t)slot_pos[s]];
                used[(size_t)slot_pos[s]] = 1;
            }
            rebuildable[m] = once ? 1 : 0;
        }
        float4 *rec = &recs[(size_t)m * pt::MESH_REC_F4];
        const ptrt_bvh_node &rn = M.nodes[0];
        rec[0] = f4(rn.bmin.x, rn.bmin.y, rn.bmin.z, as_f(root));
        rec[1] = f4(rn.bmax.x, rn.bmax.y, rn.bmax.z, as_f(M.has_transform ? 1 : 0));
        for (int r = 0; r < 3; ++r) {
            rec[2 + r] = f4(M.inverse[r * 4], M.inverse[r * 4 + 1], M.inverse[r * 4 + 2], M.inverse[r * 4 + 3]);
            rec[5 + r] = f4(M.world[r * 4], M.world[r * 4 + 1], M.world[r * 4 + 2], M.world[r * 4 + 3]);
            rec[8 + r] = f4(M.normal[r * 4], M.normal[r * 4 + 1], M.normal[r * 4 + 2], 0.0f);
        }
    }
    // TLAS: validated before anything of the old scene is freed, uploaded below
    if (int rc = upload_tlas(c, mesh_count, tlas_nodes, tlas_node_count, tlas_mesh_indices, tlas_index_count, true))
        return rc;

    free_scene(c);
    // nothing of the old scene is left: until the last upload below has succeeded the context has no geometry,
    // so a failure in between (hipMalloc) leaves it answering PTRT_E_NOT_READY instead of launching on NULL arenas
    c->have_geometry = false;
    c->n_slots = c->n_leaves = 0;
    c->n_meshes = mesh_count;
    c->h_mesh_recs = recs;
    if (int rc = push_mesh_recs(c, true))
        return rc;
    {
        // inner nodes grouped by depth (deepest last) for the level-by-level refit
        node_dst.resize(R.nodes.size() / 4, 0);
        node_depth.resize(R.nodes.size() / 4, 0);
        int maxd = 0;
        for (int d : node_depth)
            if (d > maxd)
                maxd = d;
        c->level_offset.assign((size_t)maxd + 1, 0);
        for (int d : node_depth)
            if (d >= 1)
                c->level_offset[d]++;
        // level_offset[d] currently holds the count of depth d (index 0 unused); prefix-sum it
        int run = 0;
        for (int d = 1; d <= maxd; ++d) {
            const int n = c->level_offset[d];
            c->level_offset[d - 1] = run;
            run += n;
        }
        c->level_offset[maxd] = run;
        std::vector<int> fill(c->level_offset.begin(), c->level_offset.end()), level_nodes((size_t)run);
        for (int i = 0; i < (int)node_depth.size(); ++i)
            if (node_depth[i] >= 1)
                level_nodes[(size_t)fill[node_depth[i] - 1]++] = i;
        if (int rc = upload(c, c->d_verts, all_verts))
            return rc;
        if (int rc = upload(c, c->d_slot_face, slot_face))
            return rc;
        if (int rc = upload(c, c->d_leaf_dst, leaf_dst))
            return rc;
        if (int rc = upload(c, c->d_node_dst, node_dst))
            return rc;
        if (int rc = upload(c, c->d_level_nodes, level_nodes))
            return rc;
        if (int rc = upload(c, c->d_face_src, face_src))
            return rc;
        if (int rc = upload(c, c->d_slot_pos, slot_pos))
            return rc;
        c->mesh_face_base = face_base;
        c->mesh_face_count = face_count;
        c->mesh_slot_base = slot_base;
        c->mesh_slot_count = slot_count;
        c->mesh_rebuildable = rebuildable;
        c->mesh_is_soup = is_soup;
        c->mesh_vert_base = vert_base;
        c->mesh_vert_count = vert_count;
        c->n_slots = (int)slot_face.size();
        c->n_leaves = (int)R.leaves.size();
    }
    if (int rc = upload(c, c->d_nodes, R.nodes))
        return rc;
    c->n_nodes = (int)(R.nodes.size() / 4);
    dfree(c->d_nodes2);
    if (PT_TWO_LEVEL) { // (a build whose queue modes read two-level records: measured, no gain -- DESIGN.md 3.11)
        HIP_TRY(c, hipMalloc((void **)&c->d_nodes2, (size_t)(c->n_nodes > 0 ? c->n_nodes : 1) * pt::NODE2_F4 * sizeof(float4)));
        enqueue_expand_nodes(c, c->stream);
    }
    if (int rc = upload(c, c->d_leaves, R.leaves))
        return rc;
    if (int rc = upload(c, c->d_tris, R.tris))
        return rc;
    if (!R.tris.empty()) { // the packets' geometric normals, by the same device code a repack uses (pt::packet_normal)
        const int n = (int)(R.tris.size() / 3);
        hipLaunchKernelGGL(pt::tri_normals_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_tris, n);
        HIP_TRY(c, hipGetLastError());
    }
    if (int rc = upload_tlas(c, mesh_count, tlas_nodes, tlas_node_count, tlas_mesh_indices, tlas_index_count, false))
        return rc;
    c->n_geometry_uploads++;
    c->all_single_leaf = all_leaf;
    c->any_transform = false;
    for (int m = 0; m < mesh_count; ++m)
        c->any_transform = c->any_transform || meshes[m].has_transform != 0;
    c->stack_entries = R.max_depth < 1 ? 1 : R.max_depth;
    c->pair_tri_slots = (int)(R.tris.size() / 3);
    c->pair_max_leaf = 0;
    for (const int2 &lf : R.leaves)
        if (lf.y > c->pair_max_leaf)
            c->pair_max_leaf = lf.y;
    if (int rc = upload_instance_pretests(c, meshes, mesh_count))
        return rc;
    c->have_geometry = true;
    return PTRT_OK;
}

// World-space first-pass boxes of the instances (PMODE 3, pt_render.hip.h build_pairs_general).  The reference tests an
// instance's root box in ITS space with the ray transformed by the instance's inverse matrix A|t (intersection.cuh:
// 284-297, 454-463).  Here a ray is first tested against a world-space box W that contains every ray that test can
// accept: W = the bounding box of A^-1 (corner - t) over the eight corners of the local box, in double precision (A is
// whatever matrix the caller supplies -- the reference's own mat4::inverse is not always the true inverse, which is
// why W comes from A and not from the world matrix), grown by C1 + C2 |o|_1 with
//     C1 = Kc (|A^-1|_F (|t| + |box|) + |W|_inf) + 1e-6,   C2 = Kc (|A^-1|_F |A|_F + 1),   Kc = 1e-4.
// The fp32 local test can accept a ray only if the exact ray passes within eta <= ~32 * 2^-24 (|A o + t| + |box|) of
// the local box (five roundings per slab term on quantities of that size, plus the rounded direction over a path of
// that length); mapped back to world space that is at most |A^-1|_2 eta <= 2e-6 |A^-1|_F (|A|_F |o| + |t| + |box|), and
// the world test's own rounding is below 1e-6 (|W| + |o|): Kc leaves a factor of 50.  A singular or non-finite A gets
// an infinite box (every ray is a candidate: the local test decides, as before).
// `device_recs`: the mesh records as the DEVICE holds them now (root boxes moved by ptrt_refit / ptrt_build_bvh included); when
// given, the local boxes come from there and the descriptors' BVH arrays are not read.
int upload_instance_pretests(ptrt_ctx *c, const ptrt_mesh_desc *meshes, int mesh_count, const float4 *device_recs) {
    std::vector<float4> pre((size_t)mesh_count * 2, f4(-3.0e38f, -3.0e38f, -3.0e38f, 0.0f));
    double c2max = 0.0;
    const double Kc = 1e-4, BIG = 3.0e38;
    for (int m = 0; m < mesh_count; ++m) {
        const ptrt_mesh_desc &M = meshes[m];
        pre[(size_t)m * 2 + 1] = f4(3.0e38f, 3.0e38f, 3.0e38f, 0.0f);
        if (!M.has_transform || (!device_recs && (!M.nodes || M.node_count <= 0)))
            continue;
        double A[3][3], t[3];
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k)
                A[r][k] = M.inverse[r * 4 + k];
            t[r] = M.inverse[r * 4 + 3];
        }
        const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                           A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
        double nA = 0.0, nI = 0.0, I[3][3];
        bool ok = std::isfinite(det) && std::fabs(det) > 1e-30;
        if (ok) {
            I[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det;
            I[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
            I[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
            I[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
            I[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det;
            I[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
            I[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det;
            I[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
            I[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) {
                    nA += A[r][k] * A[r][k];
                    nI += I[r][k] * I[r][k];
                }
            nA = std::sqrt(nA);
            nI = std::sqrt(nI);
        }
        double lo[3], hi[3];
        if (device_recs) {
            const float4 a = device_recs[(size_t)m * pt::MESH_REC_F4], b = device_recs[(size_t)m * pt::MESH_REC_F4 + 1];
            lo[0] = a.x, lo[1] = a.y, lo[2] = a.z, hi[0] = b.x, hi[1] = b.y, hi[2] = b.z;
        } else {
            const ptrt_bvh_node &rn = M.nodes[0];
            lo[0] = rn.bmin.x, lo[1] = rn.bmin.y, lo[2] = rn.bmin.z, hi[0] = rn.bmax.x, hi[1] = rn.bmax.y, hi[2] = rn.bmax.z;
        }
        double wmin[3] = {BIG, BIG, BIG}, wmax[3] = {-BIG, -BIG, -BIG}, boxn = 0.0, tn = 0.0, wn = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double a = std::fmax(std::fabs(lo[k]), std::fabs(hi[k]));
            boxn += a * a;
            tn += t[k] * t[k];
        }
        boxn = std::sqrt(boxn);
        tn = std::sqrt(tn);
        for (int corner = 0; ok && corner < 8; ++corner) {
            const double p[3] = {((corner & 1) ? hi[0] : lo[0]) - t[0], ((corner & 2) ? hi[1] : lo[1]) - t[1],
                                 ((corner & 4) ? hi[2] : lo[2]) - t[2]};
            for (int r = 0; r < 3; ++r) {
                const double x = I[r][0] * p[0] + I[r][1] * p[1] + I[r][2] * p[2];
                ok = ok && std::isfinite(x);
                wmin[r] = std::fmin(wmin[r], x);
                wmax[r] = std::fmax(wmax[r], x);
                wn = std::fmax(wn, std::fabs(x));
            }
        }
        const double C1 = Kc * (nI * (tn + boxn) + wn) + 1e-6, C2 = Kc * (nI * nA + 1.0);
        ok = ok && std::isfinite(C1) && std::isfinite(C2) && C1 < 1e30 && C2 < 1e3 && wn < 1e30;
        if (!ok)
            continue; // infinite box: every ray is a candidate
        c2max = std::fmax(c2max, C2);
        // (outward rounding of the double results to float: one more ulp-sized step than the margin needs)
        pre[(size_t)m * 2] = f4(std::nextafterf((float)(wmin[0] - C1), -INFINITY), std::nextafterf((float)(wmin[1] - C1), -INFINITY),
                                 std::nextafterf((float)(wmin[2] - C1), -INFINITY), 0.0f);
        pre[(size_t)m * 2 + 1] = f4(std::nextafterf((float)(wmax[0] + C1), INFINITY), std::nextafterf((float)(wmax[1] + C1), INFINITY),
                                     std::nextafterf((float)(wmax[2] + C1), INFINITY), 0.0f);
    }
    c->inst_c2 = std::nextafterf((float)c2max, INFINITY);
    if (int rc = upload(c, c->d_inst_pre, pre))
        return rc;
    c->inst_pre_ok = true;
    return PTRT_OK;
}

// Instances moved, nothing else changed (Scene::updateAccelerationStructures for a mesh whose transform is dirty,
// scene.cuh:656-743): new matrices and has_transform flags into the mesh records, new TLAS; vertices, BLASes and
// triangle packets stay where they are.
int ptrt_update_instances(ptrt_ctx *c, const ptrt_mesh_desc *meshes, int mesh_count, const ptrt_bvh_node *tlas_nodes,
                          int tlas_node_count, const int32_t *tlas_mesh_indices, int tlas_index_count) {
    if (!ctx_live(c) || !meshes || !tlas_nodes || !tlas_mesh_indices || tlas_node_count <= 0 || tlas_index_count <= 0)
        return fail(c, PTRT_E_INVALID, "ptrt_update_instances: bad argument");
    if (!c->have_geometry)
        return fail(c, PTRT_E_NOT_READY, "ptrt_update_instances: geometry not uploaded");
    if (mesh_count != c->n_meshes)
        return fail(c, PTRT_E_INVALID, "ptrt_update_instances: %d meshes, %d uploaded (use ptrt_upload_geometry)", mesh_count,
                    c->n_meshes);
    if (int rc = set_device(c))
        return rc;
    if (int rc = upload_tlas(c, mesh_count, tlas_nodes, tlas_node_count, tlas_mesh_indices, tlas_index_count, true))
        return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // frames in flight still read the old records
    c->any_transform = false;
    for (int m = 0; m < mesh_count; ++m) {
        const ptrt_mesh_desc &M = meshes[m];
        float4 *rec = &c->h_mesh_recs[(size_t)m * pt::MESH_REC_F4];
        int flags;
        std::memcpy(&flags, &rec[1].w, 4);
        flags = (flags & ~1) | (M.has_transform ? 1 : 0);
        rec[1].w = as_f(flags);
        for (int r = 0; r < 3; ++r) {
            rec[2 + r] = f4(M.inverse[r * 4], M.inverse[r * 4 + 1], M.inverse[r * 4 + 2], M.inverse[r * 4 + 3]);
            rec[5 + r] = f4(M.world[r * 4], M.world[r * 4 + 1], M.world[r * 4 + 2], M.world[r * 4 + 3]);
            rec[8 + r] = f4(M.normal[r * 4], M.normal[r * 4 + 1], M.normal[r * 4 + 2], 0.0f);
        }
        c->any_transform = c->any_transform || M.has_transform != 0;
        // flags word and the nine matrix rows only: the root box (rec[0].xyz, rec[1].xyz) on the device may have
        // been moved by ptrt_refit / ptrt_build_bvh since the upload and stays as it is
        HIP_TRY(c, hipMemcpyAsync(&c->d_mesh_recs[(size_t)m * pt::MESH_REC_F4 + 1].w, &rec[1].w, 4, hipMemcpyHostToDevice,
                                  c->stream));
        HIP_TRY(c, hipMemcpyAsync(&c->d_mesh_recs[(size_t)m * pt::MESH_REC_F4 + 2], &rec[2], 9 * sizeof(float4),
                                  hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (int rc = upload_tlas(c, mesh_count, tlas_nodes, tlas_node_count, tlas_mesh_indices, tlas_index_count, false))
        return rc;
    drop_graphs(c);
    c->n_instance_updates++;
    // First-pass boxes of the instances (PMODE 3) from the root boxes the DEVICE holds: after a ptrt_refit / ptrt_build_bvh
    // the caller's descriptors may describe the tree as it was uploaded (or BVH arrays that no longer exist), and a box
    // built from a stale root would cull instances the reference's local test hits.  The descriptors' BVH arrays are not read.
    std::vector<float4> dev_recs((size_t)mesh_count * pt::MESH_REC_F4);
    HIP_TRY(c, hipMemcpyAsync(dev_recs.data(), c->d_mesh_recs, dev_recs.size() * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (int rc = upload_instance_pretests(c, meshes, mesh_count, dev_recs.data()))
        return rc;
    return PTRT_OK;
}

// test hook: how often each kind of acceleration-structure upload ran
int ptrt_debug_upload_counts(ptrt_ctx *c, int *out2) {
    if (!ctx_live(c) || !out2)
        return PTRT_E_INVALID;
    out2[0] = c->n_geometry_uploads;
    out2[1] = c->n_instance_updates;
    return PTRT_OK;
}

int ptrt_upload_materials(ptrt_ctx *c, const ptrt_materials *m) {
    if (!ctx_live(c) || !m || m->count <= 0)
        return fail(c, PTRT_E_INVALID, "ptrt_upload_materials: bad argument");
    if (!m->albedo || !m->specular || !m->metallic || !m->roughness || !m->emission || !m->ior || !m->transmission ||
        !m->transmission_roughness || !m->clearcoat || !m->clearcoat_roughness || !m->sheen || !m->sheen_tint ||
        !m->iridescence || !m->iridescence_thickness)
        return fail(c, PTRT_E_INVALID, "ptrt_upload_materials: a required array is NULL");
    if (int rc = set_device(c))
        return rc;
    std::vector<float4> recs((size_t)m->count * 6);
    c->h_shadow_skip.assign(m->count, 0);
    bool full = false;
    for (int i = 0; i < m->count; ++i) {
        float4 *r = &recs[(size_t)i * 6];
        r[0] = f4(m->albedo[i].x, m->albedo[i].y, m->albedo[i].z, m->metallic[i]);
        r[1] = f4(m->specular[i].x, m->specular[i].y, m->specular[i].z, m->roughness[i]);
        r[2] = f4(m->emission[i].x, m->emission[i].y, m->emission[i].z, m->transmission[i]);
        r[3] = f4(m->sheen_tint[i].x, m->sheen_tint[i].y, m->sheen_tint[i].z, m->ior[i]);
        r[4] = f4(m->transmission_roughness[i], m->clearcoat[i], m->clearcoat_roughness[i], m->iridescence[i]);
        r[5] = f4(m->iridescence_thickness[i], m->sheen[i], 0.0f, 0.0f);
        c->h_shadow_skip[i] = (m->transmission[i] > 0.5f) ? 1 : 0;
        // the lean shading variant drops branches that are provably dead when all of these are <= 0
        if (!(m->transmission[i] <= 0.0f) || !(m->clearcoat[i] <= 0.0f) || !(m->iridescence[i] <= 0.0f) ||
            !(m->sheen[i] <= 0.0f))
            full = true;
    }
    if (int rc = upload(c, c->d_materials, recs))
        return rc;
    c->n_materials = m->count;
    c->mats_full = full;
    c->have_materials = true;
    if (c->have_geometry)
        return push_mesh_recs(c, false);
    return PTRT_OK;
}

int ptrt_upload_lights(ptrt_ctx *c, const ptrt_light *lights, int n) {
    if (!ctx_live(c) || n < 0 || (n > 0 && !lights))
        return fail(c, PTRT_E_INVALID, "ptrt_upload_lights: bad argument");
    if (int rc = set_device(c))
        return rc;
    std::vector<float4> recs((size_t)n * 4);
    for (int i = 0; i < n; ++i) {
        const ptrt_light &l = lights[i];
        if (l.type < 0 || l.type > 2)
            return fail(c, PTRT_E_INVALID, "light %d: unknown type %d", i, l.type);
        recs[(size_t)i * 4 + 0] = f4(l.position.x, l.position.y, l.position.z, as_f(l.type));
        recs[(size_t)i * 4 + 1] = f4(l.direction.x, l.direction.y, l.direction.z, l.intensity);
        recs[(size_t)i * 4 + 2] = f4(l.color.x, l.color.y, l.color.z, l.range);
        recs[(size_t)i * 4 + 3] = f4(l.inner_cone, l.outer_cone, l.radius, 0.0f);
    }
    if (int rc = upload(c, c->d_lights, recs))
        return rc;
    c->n_lights = n;
    return PTRT_OK;
}

int ptrt_set_camera(ptrt_ctx *c, const ptrt_camera *cam) {
    if (!ctx_live(c, false) || !cam)
        return fail(c, PTRT_E_INVALID, "ptrt_set_camera: bad argument");
    auto v = [](const ptrt_vec3 &a) { return pt::f3{a.x, a.y, a.z}; };
    c->cam.origin = v(cam->origin);
    c->cam.llc = v(cam->lower_left_corner);
    c->cam.horizontal = v(cam->horizontal);
    c->cam.vertical = v(cam->vertical);
    c->cam.u = v(cam->u);
    c->cam.v = v(cam->v);
    c->cam.w = v(cam->w);
    c->cam.lens_radius = cam->lens_radius;
    return PTRT_OK;
}

int ptrt_set_sky(ptrt_ctx *c, const ptrt_vec3 *top, const ptrt_vec3 *bottom, int use_sky) {
    if (!ctx_live(c, false))
        return fail(c, PTRT_E_INVALID, "ptrt_set_sky: bad context");
    if (top)
        c->sky_top = pt::f3{top->x, top->y, top->z};
    if (bottom)
        c->sky_bottom = pt::f3{bottom->x, bottom->y, bottom->z};
    c->use_sky = use_sky ? 1 : 0;
    return PTRT_OK;
}

int ptrt_update_vertices(ptrt_ctx *c, int mesh, const float *verts, int vert_count, int on_device) {
    if (!ctx_live(c) || !verts)
        return fail(c, PTRT_E_INVALID, "ptrt_update_vertices: bad argument");
    if (!c->have_geometry)
        return fail(c, PTRT_E_NOT_READY, "ptrt_update_vertices: geometry not uploaded");
    if (mesh < 0 || mesh >= c->n_meshes || vert_count != c->mesh_vert_count[mesh])
        return fail(c, PTRT_E_INVALID, "ptrt_update_vertices: mesh %d has %d vertices, got %d (topology must not change)",
                    mesh, (mesh >= 0 && mesh < c->n_meshes) ? c->mesh_vert_count[mesh] : -1, vert_count);
    if (int rc = set_device(c))
        return rc;
    const size_t bytes = (size_t)vert_count * 12;
    float *dst = c->d_verts + (size_t)c->mesh_vert_base[mesh] * 3;
    auto device_copy = [&](const float *src) -> int { // (a kernel of our own: the runtime's blit ran this copy at 50 GB/s)
        const size_t nf = bytes / 4;
        const int vec4 = (((size_t)src | (size_t)dst) & 15u) == 0u ? 1 : 0;
        const unsigned blocks = (unsigned)std::min<size_t>((nf / (vec4 ? 4 : 1) + 255) / 256, (size_t)c->n_cus * 8);
        hipLaunchKernelGGL(pt::copy_words_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, c->stream, src, dst, nf, vec4);
        HIP_TRY(c, hipGetLastError());
        return PTRT_OK;
    };
    if (on_device)
        return device_copy(verts);
    // Host positions: the caller may reuse its buffer the moment this returns.  Waiting for the copy would mean waiting for
    // everything in front of it on the stream -- the previous frame's trace -- so the host could not prepare frame N + 1 while the
    // GPU renders frame N (the reference's updatePTScene -> commitObjectChanges() loop: 2.04 ms per fluid frame, host and GPU
    // in turn).  The positions go through one of four pinned staging buffers of the context instead and cross PCIe behind the
    // stream's work; a buffer is waited for only when it comes round again, four updates later.
    constexpr size_t STAGE_MAX = (size_t)256 << 20;
    if (bytes > STAGE_MAX) {
        HIP_TRY(c, hipMemcpyAsync(dst, verts, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return PTRT_OK;
    }
    const int k = (int)(c->stage_n++ % ptrt_ctx::STAGES);
    if (c->stage_ev[k])
        HIP_TRY(c, hipEventSynchronize(c->stage_ev[k]));
    else
        HIP_TRY(c, hipEventCreateWithFlags(&c->stage_ev[k], hipEventDisableTiming));
    // (Positions already in PINNED memory need no host copy: they cross PCIe on a stream of their own into a device staging
    // buffer -- the call waits for that transfer alone, ~0.1 ms for 4.7 MB -- and move into the arena on the context's stream.)
    hipPointerAttribute_t pa;
    if (hipPointerGetAttributes(&pa, verts) == hipSuccess && pa.type == hipMemoryTypeHost) {
        if (!c->copy_stream) {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
            HIP_TRY(c, hipEventCreateWithFlags(&c->copy_ev, hipEventDisableTiming));
        }
        if (c->d_stage_bytes[k] < bytes) {
            dfree(c->d_stage[k]);
            c->d_stage_bytes[k] = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_stage[k], bytes));
            c->d_stage_bytes[k] = bytes;
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_stage[k], verts, bytes, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipEventRecord(c->copy_ev, c->copy_stream));
        HIP_TRY(c, hipStreamSynchronize(c->copy_stream)); // the caller's buffer is its own again
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->copy_ev, 0));
        if (int rc = device_copy(c->d_stage[k]))
            return rc;
        HIP_TRY(c, hipEventRecord(c->stage_ev[k], c->stream)); // (the staging buffer is free again behind this)
        return PTRT_OK;
    }
    (void)hipGetLastError(); // (an address HIP does not know is ordinary host memory)
    if (c->stage_bytes[k] < bytes) {
        if (c->h_stage[k])
            HIP_TRY(c, hipHostFree(c->h_stage[k]));
        c->h_stage[k] = nullptr;
        c->stage_bytes[k] = 0;
        HIP_TRY(c, hipHostMalloc(&c->h_stage[k], bytes, hipHostMallocDefault));
        c->stage_bytes[k] = bytes;
    }
    std::memcpy(c->h_stage[k], verts, bytes);
    HIP_TRY(c, hipMemcpyAsync(dst, c->h_stage[k], bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(c->stage_ev[k], c->stream));
    return PTRT_OK;
}

int ptrt_refit(ptrt_ctx *c) {
    if (!ctx_live(c))
        return fail(c, PTRT_E_INVALID, "ptrt_refit: bad context");
    if (!c->have_geometry)
        return fail(c, PTRT_E_NOT_READY, "ptrt_refit: geometry not uploaded");
    if (int rc = set_device(c))
        return rc;
    c->inst_pre_ok = false; // root boxes move on the device: the instances' first-pass boxes are stale until the next upload
    return run_graphed(c, -1, [c](hipStream_t st) { return enqueue_refit(c, st); });
}

int ptrt_build_bvh(ptrt_ctx *c, int mesh) {
    if (!ctx_live(c))
        return fail(c, PTRT_E_INVALID, "ptrt_build_bvh: bad context");
    if (!c->have_geometry)
        return fail(c, PTRT_E_NOT_READY, "ptrt_build_bvh: geometry not uploaded");
    c->inst_pre_ok = false; // (as in ptrt_refit)
    if (mesh < 0 || mesh >= c->n_meshes)
        return fail(c, PTRT_E_INVALID, "ptrt_build_bvh: no mesh %d", mesh);
    if (!c->mesh_rebuildable[mesh])
        return fail(c, PTRT_E_INVALID, "ptrt_build_bvh: mesh %d's uploaded BVH does not place every face in exactly one "
                                       "leaf position; rebuild on the host and re-upload", mesh);
    if (int rc = set_device(c))
        return rc;
    const int n = c->mesh_face_count[mesh];
    const int n_waves = (n + pt::RS_WAVE_KEYS - 1) / pt::RS_WAVE_KEYS;
    if (n > c->sort_capacity) {
        for (int k = 0; k < 2; ++k) {
            dfree(c->d_sort_keys[k]);
            dfree(c->d_sort_vals[k]);
        }
        dfree(c->d_sort_hist);
        dfree(c->d_centroids);
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(c, hipMalloc((void **)&c->d_sort_keys[k], (size_t)n * 4));
            HIP_TRY(c, hipMalloc((void **)&c->d_sort_vals[k], (size_t)n * 4));
        }
        HIP_TRY(c, hipMalloc((void **)&c->d_sort_hist, (size_t)n_waves * 256 * 4 * 2)); // counts | positions
        HIP_TRY(c, hipMalloc((void **)&c->d_centroids, (size_t)n * 12));
        if (!c->d_cbounds) { // min words all-ones, max words zero; every build leaves them so again
            HIP_TRY(c, hipMalloc((void **)&c->d_cbounds, 6 * 4));
            HIP_TRY(c, hipMemsetAsync(c->d_cbounds, 0xff, 12, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->d_cbounds + 3, 0, 12, c->stream));
        }
        c->sort_capacity = n;
        drop_graphs(c); // captured launches hold the old scratch pointers
    }
    return run_graphed(c, mesh, [c, mesh](hipStream_t st) {
        if (int rc = enqueue_build(c, mesh, st))
            return rc;
        return enqueue_refit(c, st);
    });
}

int ptrt_update_triangles(ptrt_ctx *c, int mesh, const float *verts9, int tri_count, int on_device) {
    if (!ctx_live(c) || (!verts9 && tri_count > 0) || tri_count < 0)
        return fail(c, PTRT_E_INVALID, "ptrt_update_triangles: bad argument");
    if (!c->have_geometry)
        return fail(c, PTRT_E_NOT_READY, "ptrt_update_triangles: geometry not uploaded");
    if (mesh < 0 || mesh >= c->n_meshes)
        return fail(c, PTRT_E_INVALID, "ptrt_update_triangles: no mesh %d", mesh);
    if (!c->mesh_is_soup[mesh])
        return fail(c, PTRT_E_INVALID, "ptrt_update_triangles: mesh %d is not a triangle soup (face i = vertices 3i..3i+2)", mesh);
    if (tri_count > c->mesh_face_count[mesh])
        return fail(c, PTRT_E_INVALID, "ptrt_update_triangles: mesh %d was uploaded with room for %d triangles, got %d", mesh,
                    c->mesh_face_count[mesh], tri_count);
    if (int rc = set_device(c))
        return rc;
    float *dst = c->d_verts + (size_t)c->mesh_vert_base[mesh] * 3;
    if (tri_count > 0)
        HIP_TRY(c, hipMemcpyAsync(dst, verts9, (size_t)tri_count * 36, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                  c->stream));
    const int total = c->mesh_vert_count[mesh], real = tri_count * 3;
    if (total > real)
        hipLaunchKernelGGL(pt::pad_soup_kernel, dim3((total - real + 255) / 256), dim3(256), 0, c->stream, dst, real, total);
    HIP_TRY(c, hipGetLastError());
    if (!on_device)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PTRT_OK;
}

int ptrt_read_prim_order(ptrt_ctx *c, int mesh, int32_t *out, int count) {
    if (!ctx_live(c) || !out)
        return fail(c, PTRT_E_INVALID, "ptrt_read_prim_order: bad argument");
    if (!c->have_geometry)
        return fail(c, PTRT_E_NOT_READY, "ptrt_read_prim_order: geometry not uploaded");
    if (mesh < 0 || mesh >= c->n_meshes || !c->mesh_rebuildable[mesh] || count != c->mesh_face_count[mesh])
        return fail(c, PTRT_E_INVALID, "ptrt_read_prim_order: mesh %d has %d rebuildable prim positions, asked for %d", mesh,
                    (mesh >= 0 && mesh < c->n_meshes && c->mesh_rebuildable[mesh]) ? c->mesh_face_count[mesh] : 0, count);
    if (int rc = set_device(c))
        return rc;
    std::vector<int4> sf((size_t)count);
    std::vector<int> pos((size_t)count);
    HIP_TRY(c, hipMemcpyAsync(sf.data(), c->d_slot_face + c->mesh_slot_base[mesh], (size_t)count * 16, hipMemcpyDeviceToHost,
                              c->stream));
    HIP_TRY(c, hipMemcpyAsync(pos.data(), c->d_slot_pos + c->mesh_slot_base[mesh], (size_t)count * 4, hipMemcpyDeviceToHost,
                              c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < count; ++i)
        out[pos[(size_t)i]] = sf[(size_t)i].w;
    return PTRT_OK;
}

int ptrt_upload_scene(ptrt_ctx *c, const ptrt_scene_desc *s) {
    if (!ctx_live(c) || !s)
        return fail(c, PTRT_E_INVALID, "ptrt_upload_scene: bad argument");
    if (int rc = ptrt_upload_geometry(c, s->meshes, s->mesh_count, s->tlas_nodes, s->tlas_node_count,
                                      s->tlas_mesh_indices, s->tlas_index_count))
        return rc;
    if (int rc = ptrt_upload_materials(c, &s->materials))
        return rc;
    if (int rc = ptrt_upload_lights(c, s->lights, s->light_count))
        return rc;
    if (int rc = ptrt_set_camera(c, &s->camera))
        return rc;
    if (int rc = ptrt_set_env_map(c, s->env_rgba, s->env_width, s->env_height))
        return rc;
    return ptrt_set_sky(c, &s->sky_top, &s->sky_bottom, s->use_sky);
}

int ptrt_set_env_map(ptrt_ctx *c, const float *rgba, int width, int height) {
    if (!ctx_live(c))
        return fail(c, PTRT_E_INVALID, "ptrt_set_env_map: bad context");
    if (rgba && (width < 1 || height < 1))
        return fail(c, PTRT_E_INVALID, "ptrt_set_env_map: %dx%d is not a map size", width, height);
    if (int rc = set_device(c))
        return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // frames in flight may still read the old map
    dfree(c->d_env);
    c->env_w = c->env_h = 0;
    if (!rgba)
        return PTRT_OK;
    const size_t bytes = (size_t)width * height * 16;
    HIP_TRY(c, hipMalloc((void **)&c->d_env, bytes));
    HIP_TRY(c, hipMemcpyAsync(c->d_env, rgba, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->env_w = width;
    c->env_h = height;
    return PTRT_OK;
}

void ptrt_denoiser_default_settings(ptrt_denoiser_settings *s) {
    if (!s)
        return;
    // DenoiserSettings defaults, diffuse_* channel (denoiser.cuh:40-72)
    *s = ptrt_denoiser_settings{0.06f, 0.05f, 32.0f, 4.0f, 64.0f, 0.5f, 5,    1.2f, 3.0f,
                                0.1f,  0.005f, 0.95f, 1e9f, 0.01f, 0.95f, 1,  1};
}

int ptrt_denoiser_enable(ptrt_ctx *c, const ptrt_denoiser_settings *s) {
    static_assert(sizeof(ptrt_denoiser_settings) == sizeof(pt::DenoiseSettings), "settings mirror");
    if (!ctx_live(c))
        return fail(c, PTRT_E_INVALID, "ptrt_denoiser_enable: bad context");
    if (c->rows != c->H || c->y0 != 0)
        return fail(c, PTRT_E_INVALID, "ptrt_denoiser_enable: the denoiser needs a full-frame context (its filters "
                                       "read across band borders); denoise on the presenting rank instead");
    if (int rc = set_device(c))
        return rc;
    ptrt_denoiser_settings d;
    ptrt_denoiser_default_settings(&d);
    if (s)
        d = *s;
    free_denoiser(c);
    std::memcpy(&c->dn, &d, sizeof d);
    const size_t n = c->rpix(); // "(re)create at current render resolution" (scene.cuh:1984-1996)
    HIP_TRY(c, hipMalloc((void **)&c->dn_cur4, n * 16));
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(c, hipMalloc((void **)&c->dn_c4[k], n * 16));
        HIP_TRY(c, hipMalloc((void **)&c->dn_g4[k], n * 16));
        HIP_TRY(c, hipMalloc((void **)&c->dn_h1[k], n * 16));
        HIP_TRY(c, hipMalloc((void **)&c->dn_h2[k], n * 16));
    }
    HIP_TRY(c, hipMalloc((void **)&c->dn_hobj, n * 4));
    HIP_TRY(c, hipMalloc((void **)&c->dn_motion, n * 8));
    HIP_TRY(c, hipMalloc((void **)&c->dn_out, n * 12));
    HIP_TRY(c, hipMalloc((void **)&c->dn_pvp, 16 * sizeof(float)));
    HIP_TRY(c, hipMemsetAsync(c->dn_out, 0, n * 12, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dn_motion, 0, n * 8, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->dn_cur = 0;
    c->dn_first = true;
    c->dn_on = true;
    return PTRT_OK;
}

int ptrt_denoiser_disable(ptrt_ctx *c) {
    if (!ctx_live(c))
        return fail(c, PTRT_E_INVALID, "ptrt_denoiser_disable: bad context");
    if (int rc = set_device(c))
        return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_denoiser(c);
    return PTRT_OK;
}

int ptrt_set_prev_view_proj(ptrt_ctx *c, const float *m16) {
    if (!ctx_live(c, false) || !m16)
        return fail(c, PTRT_E_INVALID, "ptrt_set_prev_view_proj: bad argument");
    std::memcpy(c->prev_view_proj, m16, sizeof c->prev_view_proj);
    return PTRT_OK;
}

int ptrt_set_render_size(ptrt_ctx *c, int render_w, int render_h) {
    if (!ctx_live(c))
        return fail(c, PTRT_E_INVALID, "ptrt_set_render_size: bad context");
    if (render_w == c->rw && render_h == c->rh)
        return PTRT_OK;
    if (c->rows != c->H || c->y0 != 0)
        return fail(c, PTRT_E_INVALID, "ptrt_set_render_size: only full-frame contexts can render at a reduced size");
    if (render_w < 1 || render_h < 1 || render_w > c->W || render_h > c->H)
        return fail(c, PTRT_E_INVALID, "ptrt_set_render_size: %dx%d is not within 1x1 .. %dx%d", render_w, render_h, c->W, c->H);
    if (c->bloom_on && (render_w < 64 || render_h < 64))
        return fail(c, PTRT_E_INVALID, "ptrt_set_render_size: bloom needs at least 64x64 (six mip levels)");
    if (int rc = set_device(c))
        return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    dfree(c->s_accum);
    dfree(c->s_normal);
    dfree(c->s_depth);
    dfree(c->s_object_id);
    free_denoiser(c); // its images have the old size; the caller re-enables it (Scene::updateScaledBuffers does)
    c->rw = render_w;
    c->rh = render_h;
    if (c->scaled()) {
        const size_t n = c->rpix();
        HIP_TRY(c, hipMalloc((void **)&c->s_accum, n * 12));
        HIP_TRY(c, hipMalloc((void **)&c->s_normal, n * 12));
        HIP_TRY(c, hipMalloc((void **)&c->s_depth, n * 4));
        HIP_TRY(c, hipMalloc((void **)&c->s_object_id, n * 4));
        HIP_TRY(c, hipMemsetAsync(c->s_accum, 0, n * 12, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return PTRT_OK;
}

int ptrt_set_bloom(ptrt_ctx *c, int enabled) {
    if (!ctx_live(c, false))
        return fail(c, PTRT_E_INVALID, "ptrt_set_bloom: bad context");
    if (!enabled) {
        c->bloom_on = 0; // the mips stay allocated, as in the reference
        return PTRT_OK;
    }
    if (c->rows != c->H || c->y0 != 0)
        return fail(c, PTRT_E_INVALID, "ptrt_set_bloom: bloom needs a full-frame context (its blur reads across band "
                                       "borders); apply it on the presenting rank instead");
    if (c->rw < 64 || c->rh < 64)
        return fail(c, PTRT_E_INVALID, "ptrt_set_bloom: bloom needs at least 64x64 pixels: below that one of the six mip "
                                       "levels is empty (the reference would read a NULL mip, scene.cuh:812-816,1175)");
    if (int rc = set_device(c))
        return rc;
    int mw = c->W, mh = c->H;
    for (int i = 0; i < 6; ++i) { // scene.cuh:809-822: sized from the FULL frame
        mw /= 2;
        mh /= 2;
        if (!c->bl_mip[i]) {
            c->touched = true; // (the first time only: a caller that sets the flag every frame keeps its frames overlapping)
            HIP_TRY(c, hipMalloc((void **)&c->bl_mip[i], (size_t)mw * mh * 12));
        }
    }
    c->bloom_on = 1;
    return PTRT_OK;
}

} // extern "C"

namespace {

// ---- ptrt_render, step by step ----------------------------------------------------------------------------------------
// Which of the two exact shapes of the queue modes' loop this frame runs (option "merged"); true while the choice is being
// sampled (the frame's launch is then timed and ordered behind the stream).
bool choose_loop_shape(ptrt_ctx *c, int spp, int max_depth, int geom) {
    // merged = -1: the two exact shapes of the queue modes' loop (shadow rays in their own traversal, or riding with the next
    // extension rays) take turns over a scene's frames 4-7 (the first four warm the clocks up); their kernel times (the event
    // ring) decide the rest at frame 8, which waits for frame 7 once
    bool tuning = false;
    const bool merged_possible = pair_mode(c, geom, true) == 4; // (only the queue mode over BLASes has the merged shape)
    c->last_merged_possible = merged_possible;
    c->merged_eff = c->merged > 0 ? 1 : 0;
    bool capturing = false; // (a caller recording this stream into a hipGraph: no host wait, no choice -- the default shape)
    if (c->merged < 0 && merged_possible && c->tune_choice < 0) { // (asked only while the choice is open: a driver call per frame otherwise)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c->stream, &cs) == hipSuccess)
            capturing = cs != hipStreamCaptureStatusNone;
        else
            (void)hipGetLastError();
    }
    if (c->merged < 0 && merged_possible && !capturing) {
        const unsigned long long key = ((((unsigned long long)c->n_geometry_uploads * 131 + (unsigned)spp) * 131 + (unsigned)max_depth) * 131 +
                                        (unsigned)(c->steal * 64 + c->fetch_min + c->csteal * 4096 + c->csteal_leaf_min * 65536)) * 131 + (unsigned)(c->leaf_min * 8 + c->leaf_pairs * 4 + c->lds_nodes * 2 + c->pair_trace);
        if (key != c->tune_key) {
            c->tune_key = key;
            c->tune_n = 0;
            c->tune_choice = -1;
        }
        if (c->tune_choice < 0 && c->tune_n < TUNE_WARM + 2 * TUNE_SAMPLES) { // warm-up frames, then merged / separate in turns
            c->merged_eff = c->tune_n >= TUNE_WARM && !((c->tune_n - TUNE_WARM) & 1);
            tuning = true;
        } else if (c->tune_choice < 0) { // the frame after the last sample: ONE host wait for that sample, once per scene and setting
            float t[2 * TUNE_SAMPLES] = {};
            bool ok = c->launches - c->tune_launch[0] < (unsigned long long)EV_RING - 8 &&
                      hipEventSynchronize(c->ev_ring[2 * (c->tune_launch[2 * TUNE_SAMPLES - 1] % EV_RING) + 1]) == hipSuccess;
            for (int i = 0; i < 2 * TUNE_SAMPLES && ok; ++i)
                ok = hipEventElapsedTime(&t[i], c->ev_ring[2 * (c->tune_launch[i] % EV_RING)], c->ev_ring[2 * (c->tune_launch[i] % EV_RING) + 1]) == hipSuccess;
            (void)hipGetLastError();
            float tm[TUNE_SAMPLES], ts[TUNE_SAMPLES];
            for (int i = 0; i < TUNE_SAMPLES; ++i) {
                tm[i] = t[2 * i];
                ts[i] = t[2 * i + 1];
            }
            std::sort(tm, tm + TUNE_SAMPLES);
            std::sort(ts, ts + TUNE_SAMPLES);
            // the separate-phase loop is the default; the merged one must be faster by TUNE_MIN_GAIN in the medians to replace it
            c->tune_choice = (ok && tm[TUNE_SAMPLES / 2] < ts[TUNE_SAMPLES / 2] * (1.0f - TUNE_MIN_GAIN)) ? 1 : 0;
            c->merged_eff = c->tune_choice;
            if (getenv("PTRT_DEBUG_LDS"))
                fprintf(stderr, "ptrt: merged loop median %.3f ms, separate %.3f ms -> %s\n", tm[TUNE_SAMPLES / 2], ts[TUNE_SAMPLES / 2],
                        c->tune_choice ? "merged" : "separate");
        } else {
            c->merged_eff = c->tune_choice;
        }
    }
    return tuning;
}

// PMODE 1: LDS layout of a workgroup (pt::carve_pm1) -> its size in bytes; fills K.lds_* and the tiles per workgroup.
size_t pm1_layout(ptrt_ctx *c, pt::KParams &K, bool full, int grid, int &pm1_wg) {
    size_t lds_main = 0;
    pm1_wg = 1;
    // PMODE 1 (pt::carve_pm1): the read-only copies are per workgroup, the lists per wave.  With the simple materials the
    // kernel exists for one tile per workgroup at five waves per SIMD and for TWO tiles at six (option pm1_wg: 0 = the larger
    // one if the scene fits its LDS budget, 1 / 2 force); the shading inputs are staged piece by piece while the workgroup
    // stays within the budget of the occupancy its kernel is built for.
    {
        const size_t shared0 = (size_t)c->pair_tri_slots * 48 + (size_t)c->pair_meshes * (pt::PAIR_PAD * 16 + 16 + 32);
        const size_t lights = (size_t)c->n_lights * 64, mats = (size_t)c->pair_meshes * (full ? 96 : 48);
        // (the lanes' blue-noise slots, 512 bytes per wave, are only worth their LDS while [A] runs in every iteration: with the
        // samples in step it runs once per sample for the whole wave and reads the table itself -- room for the materials)
        const bool stage_bn = K.sample_sync == 0;
        auto layout = [&](int wg, size_t budget, pt::KParams &P, bool may_stage = true) -> size_t { // bytes of the workgroup, 0 if over budget
            const size_t wave0 = 512 + pt::pm1_pair_bytes(c->pair_meshes) + 16;
            size_t shared = (shared0 + 15) & ~(size_t)15;
            int flags = 0;
            if (shared + wg * wave0 + (size_t)c->lds_pad > budget)
                return 0;
            const size_t extra_at = shared;
            size_t wave = wave0;
            if (may_stage && c->stage && shared + 128 + wg * (wave0 + (stage_bn ? 512 : 0)) + (size_t)c->lds_pad <= budget) {
                flags = 4 | (stage_bn ? 8 : 0);
                shared += 128;
                wave += stage_bn ? 512 : 0;
                if ((c->stage & 1) && c->n_lights > 0 && c->n_lights <= pt::LDS_LIGHTS && shared + lights + wg * wave + (size_t)c->lds_pad <= budget) {
                    flags |= 1;
                    shared += lights;
                }
                if ((c->stage & 2) && c->pair_meshes <= 42 && shared + mats + wg * wave + (size_t)c->lds_pad <= budget) {
                    flags |= 2;
                    shared += mats;
                }
            }
            P.lds_extra = (int)extra_at;
            P.lds_flags = flags;
            P.lds_wave = (int)shared;
            P.lds_wave_bytes = (int)wave;
            return shared + wg * wave + (size_t)c->lds_pad;
        };
        size_t need = 0;
        if (!full && c->pm1_wg != 1)
            need = layout(2, (size_t)pt::lds_per_workgroup(1, false, 2), K);
        if (need) {
            pm1_wg = 2;
        } else {
            need = layout(1, (size_t)pt::lds_per_wave(1, full), K);
            if (!need) // (over the budget of the kernel's occupancy: pair_mode admitted the scene, so it runs, with nothing staged)
                need = layout(1, 64 * 1024, K, false);
        }
        lds_main = need;
        K.n_tiles = grid;
    }
    return lds_main;
}

// Whether this frame may overlap its predecessor on the device, and the waits that make it safe.  Sets c->split_eff /
// c->pipelined_last, swaps in the second HDR / G-buffer set for a frame with a post chain, records the stream's head for the
// next frame.  `splittable`: the frame's kernel can be dealt to several launches at all; `recording`: the caller is capturing
// the stream into a hipGraph.
struct OverlapPlan {
    bool splittable = false, recording = false;
};
int plan_overlap(ptrt_ctx *c, pt::KParams &K, int spp, int max_depth, int pmode, int pm1_wg, int tiles_y, bool tuning, bool scaled, bool post,
                 void *out_rgb8, int out_is_device, OverlapPlan &plan) {
    // Options "split" / "pipeline": frame pipelining.  A frame ends with a tail -- its last waves drain while most of the chip
    // idles, then the next launch ramps up: ~8 % of a 1080p Cornell frame.  With the frame's rows of tiles dealt to `split`
    // launches on auxiliary streams, launch i of frame N + 1 touches the same pixels (generator states, accumulators, image
    // rows) as launch i of frame N and nothing else of that frame, so it only has to follow THAT launch, which it does on
    // its stream; it need not wait for the stream the caller sees, onto which every frame is joined by events.  That is safe
    // only while nothing else has a claim on what it reads or overwrites: no entry point that could have enqueued device work
    // or changed device data since the last frame (`touched`, set by ctx_live), no reduced render size (a post chain at full
    // size gets a second set of HDR image and G-buffers: below), no pointers into the context's buffers in the caller's hands, no loop-shape sampling (it times launches), not
    // while the caller records the stream into a graph, and
    // a DEVICE target other than the previous frame's (whatever consumes that one on the stream is still entitled to it).
    // A frame that cannot overlap is ONE launch on the context's stream, as ever (concurrent launches of one frame buy
    // nothing: Cornell 1.85 vs 1.82 ms) -- followed by an event the next frame's launches wait for if that one can.
    // What an overlapping frame DOES wait for, besides its predecessor's launches: everything that was on the stream when the
    // PREVIOUS ptrt_render was called -- the consumers of the frame before that one, whose target a double-buffering caller
    // hands in again now.  (head_ev alternates: [head_n & 1] is recorded now, the other one is the previous call's.)
    c->split_eff = 1;
    c->pipelined_last = false;
    const int n_split = c->split < ptrt_ctx::MAX_SPLIT ? c->split : ptrt_ctx::MAX_SPLIT;
    const bool splittable = c->pipeline && n_split > 1 && tiles_y >= 2 * n_split && !async_applicable(c) &&
                            !wavefront_applicable(c, spp, max_depth) && !(pmode == 1 && pm1_wg == 2) &&
                            !(pmode == 2 && c->lds_nodes && c->stack_entries > 0);
    // (a caller recording the stream into a hipGraph: the bookkeeping events below would become nodes of its graph, and a replay
    // never passes through here -- such a frame is one ordered launch, and so is the first frame after it)
    bool recording = false;
    if (splittable) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c->stream, &cs) != hipSuccess)
            (void)hipGetLastError();
        recording = cs != hipStreamCaptureStatusNone;
    }
    // (the previous frame's post chain reads the HDR image and G-buffers this frame's trace would overwrite: a frame WITH a chain
    // writes the other set, below; one without -- the chain was switched off in between -- waits for the stream instead.  A
    // changed number of launches moves the rows between the auxiliary streams: only a one-launch frame may precede it.)
    if (splittable && !recording && !c->touched && !c->escaped && !tuning && !scaled && out_rgb8 && out_is_device &&
        out_rgb8 != c->prev_out && c->prev_stream == c->stream && (c->prev_split == n_split || c->prev_split == 1) &&
        !(c->prev_post && !post)) {
        {
            // (the second set of HDR image and G-buffers: all four or none -- a failed allocation leaves the frame unpipelined)
            bool alt_ok = true;
            if (post && !c->alt_accum) {
                float *a = nullptr, *nrm = nullptr, *d = nullptr;
                int *o = nullptr;
                alt_ok = hipMalloc((void **)&a, c->npix * 3 * sizeof(float)) == hipSuccess &&
                         hipMalloc((void **)&nrm, c->npix * 3 * sizeof(float)) == hipSuccess &&
                         hipMalloc((void **)&d, c->npix * sizeof(float)) == hipSuccess &&
                         hipMalloc((void **)&o, c->npix * sizeof(int)) == hipSuccess;
                if (alt_ok) {
                    c->alt_accum = a;
                    c->alt_normal = nrm;
                    c->alt_depth = d;
                    c->alt_object_id = o;
                } else {
                    (void)hipGetLastError();
                    dfree(a);
                    dfree(nrm);
                    dfree(d);
                    dfree(o);
                }
            }
            if (alt_ok) {
            for (int i = 0; i < n_split; ++i)
                if (!c->aux_stream[i]) {
                    HIP_TRY(c, hipStreamCreateWithFlags(&c->aux_stream[i], hipStreamNonBlocking));
                    HIP_TRY(c, hipEventCreateWithFlags(&c->split_join[i], hipEventDisableTiming));
                }
            for (int i = 0; i < n_split; ++i) {
                if (c->prev_split != n_split) // the previous frame was one launch on the stream: follow it (and only it)
                    HIP_TRY(c, hipStreamWaitEvent(c->aux_stream[i], c->split_fork, 0));
                if (c->head_ev[(c->head_n + 1) & 1])
                    HIP_TRY(c, hipStreamWaitEvent(c->aux_stream[i], c->head_ev[(c->head_n + 1) & 1], 0));
            }
            if (post) {
                // the post chain of the previous frame may still be reading the HDR image and the G-buffers on the stream:
                // this frame's trace writes the OTHER set (its own post chain, enqueued behind the join, reads that one)
                std::swap(c->d_accum, c->alt_accum);
                std::swap(c->d_normal, c->alt_normal);
                std::swap(c->d_depth, c->alt_depth);
                std::swap(c->d_object_id, c->alt_object_id);
                K.accum = c->d_accum;
                K.normal = c->d_normal;
                K.depth = c->d_depth;
                K.object_id = c->d_object_id;
            }
            c->split_eff = n_split;
            c->pipelined_last = true;
            }
        }
    }
    if (splittable && !recording) { // the stream's head at this call, for the NEXT frame
        hipEvent_t &he = c->head_ev[c->head_n & 1];
        if (!he)
            HIP_TRY(c, hipEventCreateWithFlags(&he, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(he, c->stream));
        ++c->head_n;
    }
    plan.splittable = splittable;
    plan.recording = recording;
    return PTRT_OK;
}

// The frame's trace launch(es) in the loop shape chosen for the scene.
int launch_frame(ptrt_ctx *c, pt::KParams &K, bool full, int spp, int max_depth, int geom, int pmode, int pm1_wg, int grid, size_t lds,
                 size_t lds_main) {
    if (async_applicable(c)) {
        if (int rc = run_async(c, K, full))
            return rc;
        c->last_mode = 2;
    } else if (wavefront_applicable(c, spp, max_depth)) {
        if (int rc = run_wavefront(c, K, full, spp, max_depth))
            return rc;
        c->last_mode = 1;
    } else if (pmode == 1 && pm1_wg == 2) {
        c->split_eff = 1;
        hipLaunchKernelGGL((pt::path_trace_kernel<0, false, 1, 2>), dim3((grid + 1) / 2), dim3(128), lds_main, c->stream, K);
    }
    else if (pmode == 1)
        { if (int rc = launch_trace<0, 1>(c, K, full, grid, lds_main)) return rc; }
    else if (pmode == 4)
        { if (int rc = launch_trace<1, 4>(c, K, full, grid, lds_main)) return rc; }
    else if (pmode == 2 && c->lds_nodes && c->stack_entries > 0) {
        // four tiles per workgroup, one LDS copy of the mesh heads and of the BLAS top levels (north star: "BVH nodes
        // ... staged in LDS")
        const size_t lds4 = (size_t)c->pair_meshes * (32 + pt::TOP_NODES * 64) +
                            4 * (512 + (size_t)c->pair_meshes * 128 + 256 + (size_t)c->stack_entries * 512 + pt::LEAF_PAIR_BYTES);
        if (lds4 > 64 * 1024)
            return fail(c, PTRT_E_INVALID, "lds_nodes: %zu bytes of LDS per workgroup", lds4);
        K.n_tiles = grid;
        K.top_off = c->lds_nodes == 2 ? 1 : 0;
        if (full)
            hipLaunchKernelGGL((pt::path_trace_kernel<1, true, 2, 4>), dim3((grid + 3) / 4), dim3(256), lds4, c->stream, K);
        else
            hipLaunchKernelGGL((pt::path_trace_kernel<1, false, 2, 4>), dim3((grid + 3) / 4), dim3(256), lds4, c->stream, K);
    } else if (pmode == 2)
        { if (int rc = launch_trace<1, 2>(c, K, full, grid, lds_main)) return rc; }
    else if (pmode == 3)
        { if (int rc = launch_trace<2, 3>(c, K, full, grid, lds_main)) return rc; }
    else if (geom == 0)
        { if (int rc = launch_trace<0, 0>(c, K, full, grid, lds)) return rc; }
    else if (geom == 1)
        { if (int rc = launch_trace<1, 0>(c, K, full, grid, lds)) return rc; }
    else
        { if (int rc = launch_trace<2, 0>(c, K, full, grid, lds)) return rc; }
    HIP_TRY(c, hipGetLastError());
    return PTRT_OK;
}

// Joins a split frame onto the context's stream and notes what the NEXT frame may follow.
int join_frame(ptrt_ctx *c, const OverlapPlan &plan, bool post) {
    const bool splittable = plan.splittable, recording = plan.recording;
    for (int i = 0; c->split_eff > 1 && i < c->split_eff; ++i) { // join: what follows on the context's stream follows every launch
        HIP_TRY(c, hipEventRecord(c->split_join[i], c->aux_stream[i]));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->split_join[i], 0));
    }
    c->prev_split = 0;
    c->prev_post = post;
    if (recording) {
        c->touched = true; // (what a replayed graph does to the buffers is not this call's to know: the next frame waits for the stream)
    } else if (c->split_eff > 1) {
        c->prev_split = c->split_eff;
    } else if (splittable && c->last_mode == 0) { // one launch on the stream: the point the next frame's launches may follow
        if (!c->split_fork)
            HIP_TRY(c, hipEventCreateWithFlags(&c->split_fork, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->split_fork, c->stream));
        c->prev_split = 1;
    }
    return PTRT_OK;
}

} // namespace

extern "C" {

int ptrt_render(ptrt_ctx *c, int frame_index, int spp, int max_depth, void *out_rgb8, int out_is_device) {
    if (!ctx_live(c, false))
        return fail(c, PTRT_E_INVALID, "ptrt_render: bad context");
    if (!c->have_geometry || !c->have_materials)
        return fail(c, PTRT_E_NOT_READY, "ptrt_render: %s not uploaded", c->have_geometry ? "materials" : "geometry");
    if (c->n_materials < c->n_meshes)
        return fail(c, PTRT_E_NOT_READY, "ptrt_render: %d materials for %d meshes", c->n_materials, c->n_meshes);
    if (!c->rng_ready)
        return fail(c, PTRT_E_NOT_READY, "ptrt_render: generator states not initialised (ptrt_reset_rng)");
    if (spp < 1 || max_depth < 1 || spp > 32767 || max_depth > 32767) // (a lane keeps its sample index and bounce in one register's halves)
        return fail(c, PTRT_E_INVALID, "ptrt_render: spp=%d max_depth=%d (1..32767)", spp, max_depth);
    if (frame_index < 0 || frame_index > INT_MAX - spp) // (sample s of the frame indexes the jitter table with (frame_index + s) % 16)
        return fail(c, PTRT_E_INVALID, "ptrt_render: frame_index=%d", frame_index);
    if (int rc = set_device(c))
        return rc;
    pt::KParams K = make_params(c);
    K.spp = spp;
    K.max_depth = max_depth;
    // Samples in step (path_trace_kernel [A], K.sample_sync): the lanes of a wave start a sample together, so a wave's lanes sit at the
    // same bounce -- a first hit samples no light and the whole wave skips [C2] / [D] in that iteration, [A] runs once per sample
    // for 64 lanes instead of every iteration for a quarter of them -- at the price of lanes that wait for the longest path of
    // the sample.  Pays while most paths run to the depth limit: Cornell 4 bounces 1.764 -> 1.638 ms (3 bounces 1.42 -> 1.26),
    // `many` 15.9 -> 14.8; loses once Russian roulette thins the wave (5 bounces 1.966 -> 1.980, 6: 2.10 -> 2.31, 8: 2.26 ->
    // 2.80); scenes of short paths are indifferent once a path's last vertex costs nothing (showcase 3.90 -> 3.90; the fluid frame
    // gains 3 %), so the depth limit alone decides.  Releasing the waiting lanes early
    // (when few are still under way, or when many wait) was measured at every threshold and is worse than both extremes.
    K.sample_sync = c->sample_sync >= 0 ? c->sample_sync : (max_depth <= 4 ? 1 : 0);
    c->sample_sync_eff = K.sample_sync;
    K.tile_run = c->tile_run;
    K.frame_count = frame_index;
    unsigned char *frame_rgb8 = (out_rgb8 && out_is_device) ? (unsigned char *)out_rgb8 : c->d_rgb8;
    // the stage that produces the final HDR image also tonemaps it; earlier stages skip theirs
    const bool scaled = c->scaled(), denoise = c->dn_on && c->dn_active, bloom = c->bloom_on != 0;
    // PTRT_OUT_DEVICE_FRAME: out_rgb8 is the whole W x H frame on this device; a band / strip context writes its rows where
    // they belong in it -- no image of its own, nothing for a tile farm to copy (ptrt_farm_*)
    const bool into_frame = out_rgb8 && out_is_device == PTRT_OUT_DEVICE_FRAME;
    if (into_frame && (denoise || bloom || scaled))
        return fail(c, PTRT_E_INVALID, "ptrt_render: PTRT_OUT_DEVICE_FRAME with the denoiser, bloom or a reduced render size");
    c->last_rgb8 = into_frame ? nullptr : frame_rgb8; // (a frame target is the caller's: ptrt_read_buffer(RGB8) has nothing to read)
    c->last_frame_target = into_frame ? out_rgb8 : nullptr;
    K.rgb8_frame = into_frame ? 1 : 0;
    K.rgb8 = (denoise || bloom || scaled) ? nullptr : frame_rgb8;
    if (c->count_rays)
        K.counters = c->d_counters;
    const int tiles_y = (K.rows + 7) / 8;
    const int grid = K.tiles_x * tiles_y;
    const int geom = pick_geom(c);
    const bool full = c->mats_full || c->force_full;
    // merged = -1: the two exact shapes of the queue modes' loop take turns over a scene's first frames and the faster one stays
    const bool tuning = choose_loop_shape(c, spp, max_depth, geom);
    const int pmode = pair_mode(c, geom, c->merged_eff != 0);
    c->last_pmode = pmode;
    // (round 2: the merged loop was at its best WITHOUT shadow-ray subtree stealing, 3.98 vs 4.17 ms on the showcase frame -- its
    // yields served ten shadow pairs at the price of sixty closest-hit walks; with the closest-hit walks stolen from as well the
    // yields pay for both kinds: 3.19 ms with, 3.49 without)
    if (c->merged < 0 && pmode == 4 && c->csteal == 0)
        K.steal = 0;
    const size_t lds = pmode ? pair_lds_bytes(c, pmode) : ((geom == 0) ? 0 : (size_t)c->stack_entries * 64 * sizeof(uint2));
    // (the heads only change with the mesh records: a frame since whose predecessor no entry point touched the device keeps them)
    if (pmode == 3 && (c->touched || !c->heads_fresh)) { // (outside the timed kernel: a 136-thread copy)
        c->heads_fresh = true;
        hipLaunchKernelGGL(pt::gather_tlas_heads_kernel, dim3((c->n_tlas_index + 63) / 64), dim3(64), 0, c->stream,
                           c->d_mesh_recs, c->d_inst_pre, c->d_tlas_mesh_ids, c->n_tlas_index, c->d_tlas_heads,
                           c->inst_pre_ok ? 1 : 0);
        HIP_TRY(c, hipGetLastError());
    }
    // PMODE 1: the workgroup's LDS layout (triangles, tables and -- while the budget of its occupancy lasts -- the shading inputs)
    size_t lds_main = lds + (size_t)c->lds_pad;
    int pm1_wg = 1;
    if (pmode == 1)
        lds_main = pm1_layout(c, K, full, grid, pm1_wg);
    if (c->launches == 0 && getenv("PTRT_DEBUG_LDS"))
        fprintf(stderr, "ptrt: pmode %d, %d meshes in the leaf, %d triangle slots, stack %d, LDS %zu + %zu bytes per workgroup\n", pmode,
                c->pair_meshes, c->pair_tri_slots, c->stack_entries, lds, lds_main - lds);
    const int slot = (int)(c->launches % EV_RING);
    const bool timing = c->time_kernels || tuning;
    if (timing)
        HIP_TRY(c, hipEventRecord(c->ev_ring[2 * slot], c->stream));
    // frame pipelining (options "split" / "pipeline"): may this frame's launches follow the previous frame's instead of the stream?
    const bool post = denoise || bloom;
    OverlapPlan plan;
    if (int rc = plan_overlap(c, K, spp, max_depth, pmode, pm1_wg, tiles_y, tuning, scaled, post, out_rgb8, out_is_device, plan))
        return rc;
    // Lane refill (launch_trace): where it was measured to pay.  Overlapping 1080p Cornell frames 1.67 -> 1.62 ms, 8 bounces 2.12
    // -> 1.88, 4K 6.65 -> 6.27; a frame alone on the chip ends in a long drain of half-empty persistent waves (1.81 -> 1.97),
    // short pixels finish before the refill pays for itself (1 spp: 0.43 -> 0.61), and beside a post chain the persistent waves
    // keep the chain's kernels waiting for a place on the chip (balanced preset 2.41 -> 2.50).
    // (And a launch must hold at least two tiles per persistent wave -- 1280 x 720 measured even, smaller frames lose.)
    const long per_launch = (long)grid / (c->split_eff > 1 ? c->split_eff : 1);
    const long resident = (long)c->n_cus * (c->persist > 0 ? c->persist : 4 * pt::waves_per_simd(1, full, 1));
    c->refill_eff = pmode == 1 && pm1_wg == 1 &&
                    (c->refill == 2 || (c->refill == 1 && c->pipelined_last && !full && !denoise && !bloom &&
                                        (long)spp * max_depth >= 16 && per_launch >= 2 * resident));
    c->prev_out = (out_rgb8 && out_is_device) ? out_rgb8 : nullptr;
    c->prev_stream = c->stream;
    c->touched = false;
    c->last_mode = 0;
    c->launch_timed[slot] = 0; // (launch_trace sets it for the launches it puts events around; the other loop shapes leave none)
    if (int rc = launch_frame(c, K, full, spp, max_depth, geom, pmode, pm1_wg, grid, lds, lds_main))
        return rc;
    if (int rc = join_frame(c, plan, post))
        return rc;
    if (timing)
        HIP_TRY(c, hipEventRecord(c->ev_ring[2 * slot + 1], c->stream));
    if (tuning) {
        if (c->tune_n >= TUNE_WARM)
            c->tune_launch[c->tune_n - TUNE_WARM] = c->launches;
        ++c->tune_n;
    }
    c->launches++;
    c->timed = timing;
    float *current = K.accum; // `current_image` of Scene::render_to_device (scene.cuh:1086)
    if (denoise) {
        if (int rc = run_denoiser(c, K, (bloom || scaled) ? nullptr : frame_rgb8))
            return rc;
        current = c->dn_out;
    }
    if (bloom) {
        if (int rc = run_bloom(c, current, c->rw, c->rh, scaled ? nullptr : frame_rgb8))
            return rc;
    }
    if (scaled) { // up-scale into the full-size colour buffer (scene.cuh:1192-1201) + tonemap
        hipLaunchKernelGGL(pt::upscale_tonemap_kernel, dim3((c->W + 63) / 64, (c->H + 3) / 4), dim3(256), 0, c->stream,
                           c->d_accum, current, c->W, c->H, c->rw, c->rh, frame_rgb8);
        HIP_TRY(c, hipGetLastError());
    }
    if (out_rgb8 && out_is_device && !into_frame) // (a frame shared by several contexts is marked by whoever joins them)
        ring_mark_rendered(out_rgb8, c->stream);
    if (out_rgb8 && !out_is_device) {
        HIP_TRY(c, hipMemcpyAsync(out_rgb8, c->d_rgb8, c->npix * 3, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return PTRT_OK;
}

// Presenting rank of the tile farm: steps 3-7 of Scene::render_to_device (motion vectors,
// denoiser, bloom, tonemap; scene.cuh:1103-1208) of a FULL-FRAME context over a frame whose HDR image and
// G-buffers were rendered elsewhere -- the band contexts -- and gathered into device memory.
int ptrt_post_frame(ptrt_ctx *c, const float *accum, const float *normal, const float *depth, const int32_t *object_id,
                    void *out_rgb8, int out_is_device) {
    if (!ctx_live(c) || !accum || !normal || !depth || !object_id)
        return fail(c, PTRT_E_INVALID, "ptrt_post_frame: bad argument");
    if (c->rows != c->H || c->y0 != 0)
        return fail(c, PTRT_E_INVALID, "ptrt_post_frame: needs a full-frame context (this one holds rows %d..%d of %d)", c->y0,
                    c->y0 + c->rows, c->H);
    if (c->scaled())
        return fail(c, PTRT_E_INVALID, "ptrt_post_frame: the frame must have the context's size (render size %dx%d != %dx%d)",
                    c->rw, c->rh, c->W, c->H);
    const bool denoise = c->dn_on && c->dn_active, bloom = c->bloom_on != 0;
    if (!denoise && !bloom)
        return fail(c, PTRT_E_INVALID, "ptrt_post_frame: neither denoiser nor bloom is enabled (gather the RGB8 bands instead)");
    if (int rc = set_device(c))
        return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_accum, accum, c->npix * 3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_normal, normal, c->npix * 3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_depth, depth, c->npix * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_object_id, object_id, c->npix * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    pt::KParams K = make_params(c);
    unsigned char *frame_rgb8 = (out_rgb8 && out_is_device) ? (unsigned char *)out_rgb8 : c->d_rgb8;
    c->last_rgb8 = frame_rgb8;
    float *current = K.accum;
    if (denoise) {
        if (int rc = run_denoiser(c, K, bloom ? nullptr : frame_rgb8))
            return rc;
        current = c->dn_out;
    }
    if (bloom) {
        if (int rc = run_bloom(c, current, c->rw, c->rh, frame_rgb8))
            return rc;
    }
    if (out_rgb8 && out_is_device)
        ring_mark_rendered(out_rgb8, c->stream);
    if (out_rgb8 && !out_is_device) {
        HIP_TRY(c, hipMemcpyAsync(out_rgb8, c->d_rgb8, c->npix * 3, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return PTRT_OK;
}

int ptrt_sync(ptrt_ctx *c) {
    if (!ctx_live(c, false))
        return fail(c, PTRT_E_INVALID, "ptrt_sync: bad context");
    if (int rc = set_device(c))
        return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PTRT_OK;
}

int ptrt_present_destroy(ptrt_ctx *c) {
    if (!ctx_live(c))
        return fail(c, PTRT_E_INVALID, "ptrt_present_destroy: bad context");
    if (c->present.empty())
        return PTRT_OK;
    if (int rc = set_device(c))
        return rc;
    (void)hipStreamSynchronize(c->stream);
    free_present(c);
    return PTRT_OK;
}

int ptrt_present_create(ptrt_ctx *c, int slots) {
    if (!ctx_live(c) || slots < 1 || slots > 8)
        return fail(c, PTRT_E_INVALID, "ptrt_present_create: 1..8 slots");
    if (int rc = ptrt_present_destroy(c))
        return rc;
    const size_t bytes = c->npix * 3;
    c->present.resize((size_t)slots);
    for (auto &s : c->present) {
        HIP_TRY(c, hipMalloc((void **)&s.dev, bytes));
        HIP_TRY(c, hipHostMalloc((void **)&s.host, bytes, hipHostMallocDefault));
        HIP_TRY(c, hipEventCreateWithFlags(&s.rendered, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    if (!c->present_stream)
        HIP_TRY(c, hipStreamCreateWithFlags(&c->present_stream, hipStreamNonBlocking));
    return PTRT_OK;
}

int ptrt_present_map(ptrt_ctx *c, int slot, void **device_pixels) {
    if (!ctx_live(c) || !device_pixels || slot < 0 || slot >= (int)c->present.size())
        return fail(c, PTRT_E_INVALID, "ptrt_present_map: no such slot (ptrt_present_create first)");
    if (int rc = set_device(c))
        return rc;
    auto &s = c->present[(size_t)slot];
    if (s.in_flight) { // the frame about to be overwritten must have reached the host
        HIP_TRY(c, hipEventSynchronize(s.done));
        s.in_flight = false;
    }
    *device_pixels = s.dev;
    return PTRT_OK;
}

int ptrt_present_unmap(ptrt_ctx *c, int slot) {
    if (!ctx_live(c) || slot < 0 || slot >= (int)c->present.size())
        return fail(c, PTRT_E_INVALID, "ptrt_present_unmap: no such slot");
    if (int rc = set_device(c))
        return rc;
    auto &s = c->present[(size_t)slot];
    // the download runs on its own stream behind an event, so it overlaps the NEXT frame's kernels
    // (a copy enqueued on the render stream would only be asynchronous to the host)
    HIP_TRY(c, hipEventRecord(s.rendered, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->present_stream, s.rendered, 0));
    HIP_TRY(c, hipMemcpyAsync(s.host, s.dev, c->npix * 3, hipMemcpyDeviceToHost, c->present_stream));
    HIP_TRY(c, hipEventRecord(s.done, c->present_stream));
    s.in_flight = true;
    return PTRT_OK;
}

int ptrt_present_acquire(ptrt_ctx *c, int slot, const unsigned char **host_pixels) {
    if (!ctx_live(c) || !host_pixels || slot < 0 || slot >= (int)c->present.size())
        return fail(c, PTRT_E_INVALID, "ptrt_present_acquire: no such slot");
    if (int rc = set_device(c))
        return rc;
    auto &s = c->present[(size_t)slot];
    if (s.in_flight) {
        HIP_TRY(c, hipEventSynchronize(s.done));
        s.in_flight = false;
    }
    *host_pixels = s.host;
    return PTRT_OK;
}

namespace {
bool ring_live(ptrt_ring *r) {
    std::lock_guard<std::mutex> lock(g_ring_mutex);
    return r && g_rings.count(r);
}
void ring_free(ptrt_ring *r) {
    (void)hipSetDevice(r->device);
    if (r->copy_stream) {
        (void)hipStreamSynchronize(r->copy_stream);
        (void)hipStreamDestroy(r->copy_stream);
    }
    for (auto &s : r->slots) {
        if (s.dev)
            (void)hipFree(s.dev);
        if (s.host)
            (void)hipHostFree(s.host);
        if (s.rendered)
            (void)hipEventDestroy(s.rendered);
        if (s.done)
            (void)hipEventDestroy(s.done);
    }
    delete r;
}
} // namespace

int ptrt_ring_create(int device, size_t frame_bytes, int slots, ptrt_ring **out) {
    if (!out)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_ring_create: out is NULL");
    *out = nullptr;
    if (frame_bytes == 0 || slots < 1 || slots > 8)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_ring_create: %zu bytes, %d slots (1..8)", frame_bytes, slots);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, PTRT_E_NO_DEVICE, "no HIP device available; this library has no CPU path");
    if (device < 0 || device >= ndev)
        return fail(nullptr, PTRT_E_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    ptrt_ring *r = new ptrt_ring;
    r->device = device;
    r->bytes = frame_bytes;
    r->slots.resize((size_t)slots);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&r->copy_stream, hipStreamNonBlocking);
    for (auto &s : r->slots) {
        if (e == hipSuccess)
            e = hipMalloc((void **)&s.dev, frame_bytes);
        if (e == hipSuccess)
            e = hipHostMalloc((void **)&s.host, frame_bytes, hipHostMallocDefault);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&s.rendered, hipEventDisableTiming);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        ring_free(r);
        return fail(nullptr, PTRT_E_HIP, "ptrt_ring_create: %s", hipGetErrorString(e));
    }
    {
        std::lock_guard<std::mutex> lock(g_ring_mutex);
        g_rings.insert(r);
    }
    *out = r;
    return PTRT_OK;
}

int ptrt_ring_map(ptrt_ring *r, int slot, void **device_pixels) {
    if (!ring_live(r) || !device_pixels || slot < 0 || slot >= (int)r->slots.size())
        return fail(nullptr, PTRT_E_INVALID, "ptrt_ring_map: bad ring or slot");
    HIP_TRY(nullptr, hipSetDevice(r->device));
    auto &s = r->slots[(size_t)slot];
    if (s.in_flight) { // the frame about to be overwritten must have reached the host
        HIP_TRY(nullptr, hipEventSynchronize(s.done));
        s.in_flight = false;
    }
    {
        std::lock_guard<std::mutex> lock(g_ring_mutex);
        s.marked = false;
    }
    *device_pixels = s.dev;
    return PTRT_OK;
}

int ptrt_ring_unmap(ptrt_ring *r, int slot) {
    if (!ring_live(r) || slot < 0 || slot >= (int)r->slots.size())
        return fail(nullptr, PTRT_E_INVALID, "ptrt_ring_unmap: bad ring or slot");
    HIP_TRY(nullptr, hipSetDevice(r->device));
    auto &s = r->slots[(size_t)slot];
    bool marked;
    {
        std::lock_guard<std::mutex> lock(g_ring_mutex);
        marked = s.marked;
    }
    // a slot not written through ptrt_render: behind everything already submitted to the device's blocking
    // streams, which is what cudaGraphicsUnmapResources guarantees the GL side
    if (!marked)
        HIP_TRY(nullptr, hipEventRecord(s.rendered, nullptr));
    HIP_TRY(nullptr, hipStreamWaitEvent(r->copy_stream, s.rendered, 0));
    HIP_TRY(nullptr, hipMemcpyAsync(s.host, s.dev, r->bytes, hipMemcpyDeviceToHost, r->copy_stream));
    HIP_TRY(nullptr, hipEventRecord(s.done, r->copy_stream));
    s.in_flight = true;
    return PTRT_OK;
}

int ptrt_ring_acquire(ptrt_ring *r, int slot, const unsigned char **host_pixels) {
    if (!ring_live(r) || !host_pixels || slot < 0 || slot >= (int)r->slots.size())
        return fail(nullptr, PTRT_E_INVALID, "ptrt_ring_acquire: bad ring or slot");
    HIP_TRY(nullptr, hipSetDevice(r->device));
    auto &s = r->slots[(size_t)slot];
    if (s.in_flight) {
        HIP_TRY(nullptr, hipEventSynchronize(s.done));
        s.in_flight = false;
    }
    *host_pixels = s.host;
    return PTRT_OK;
}

void ptrt_ring_destroy(ptrt_ring *r) {
    {
        std::lock_guard<std::mutex> lock(g_ring_mutex);
        if (!r || !g_rings.count(r))
            return;
        g_rings.erase(r);
    }
    ring_free(r);
}

int ptrt_last_kernel_ms(ptrt_ctx *c, float *trace_ms, float *tonemap_ms) {
    if (!ctx_live(c) || !c->timed)
        return fail(c, PTRT_E_NOT_READY, "ptrt_last_kernel_ms: nothing rendered yet");
    if (int rc = set_device(c))
        return rc;
    const int slot = (int)((c->launches - 1) % EV_RING);
    HIP_TRY(c, hipEventSynchronize(c->ev_ring[2 * slot + 1]));
    float ms = 0.0f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_ring[2 * slot], c->ev_ring[2 * slot + 1]));
    if (trace_ms)
        *trace_ms = ms;
    if (tonemap_ms)
        *tonemap_ms = 0.0f; // the tonemap is fused into the path-trace kernel
    return PTRT_OK;
}

int ptrt_set_stream(ptrt_ctx *c, void *hip_stream) {
    if (!ctx_live(c))
        return fail(c, PTRT_E_INVALID, "ptrt_set_stream: bad context");
    if (int rc = set_device(c))
        return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // finish what was queued on the old stream
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    c->launches = 0;
    c->timed = false;
    memset(c->launch_timed, 0, sizeof c->launch_timed);
    return PTRT_OK;
}

int ptrt_kernel_ms_history(ptrt_ctx *c, float *out_ms, int max_n) {
    if (!ctx_live(c) || !out_ms || max_n < 0)
        return fail(c, PTRT_E_INVALID, "ptrt_kernel_ms_history: bad argument");
    if (int rc = set_device(c))
        return rc;
    if (!c->time_kernels)
        return 0; // (option time_kernels = 0: no events were recorded)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long n = c->launches < (unsigned long long)EV_RING ? c->launches : EV_RING;
    if (n > (unsigned long long)max_n)
        n = max_n;
    for (unsigned long long i = 0; i < n; ++i) {
        const int slot = (int)((c->launches - n + i) % EV_RING);
        HIP_TRY(c, hipEventElapsedTime(&out_ms[i], c->ev_ring[2 * slot], c->ev_ring[2 * slot + 1]));
    }
    return (int)n;
}

// Durations of the LAUNCHES of the last frames that were dealt to the auxiliary streams with option "time_launches" on, oldest
// first: trace_ms[k] = the path-trace kernel of launch k, tail_ms[k] = from its end to the end of the tonemap pass behind it
// (lane refill; 0 otherwise).  A frame of `split` launches contributes `split` entries.  Waits for the stream.
int ptrt_launch_ms_history(ptrt_ctx *c, float *trace_ms, float *tail_ms, int max_n) {
    if (!ctx_live(c) || !trace_ms || max_n < 0)
        return fail(c, PTRT_E_INVALID, "ptrt_launch_ms_history: bad argument");
    if (int rc = set_device(c))
        return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const unsigned long long frames = c->launches < (unsigned long long)EV_RING ? c->launches : EV_RING;
    // newest first, then reversed into the caller's arrays
    std::vector<float> a, b;
    for (unsigned long long f = 0; f < frames && (int)a.size() < max_n; ++f) {
        const int slot = (int)((c->launches - 1 - f) % EV_RING);
        if (!c->launch_timed[slot])
            break; // (the run of timed frames ends here)
        for (int i = ptrt_ctx::MAX_SPLIT - 1; i >= 0 && (int)a.size() < max_n; --i)
            if ((c->launch_timed[slot] >> i) & 1) {
                float t = 0.0f, u = 0.0f;
                hipEvent_t *ev = &c->launch_ev[i][3 * slot];
                HIP_TRY(c, hipEventSynchronize(ev[2]));
                HIP_TRY(c, hipEventElapsedTime(&t, ev[0], ev[1]));
                HIP_TRY(c, hipEventElapsedTime(&u, ev[1], ev[2]));
                a.push_back(t);
                b.push_back(u);
            }
    }
    const int n = (int)a.size();
    for (int k = 0; k < n; ++k) {
        trace_ms[k] = a[n - 1 - k];
        if (tail_ms)
            tail_ms[k] = b[n - 1 - k];
    }
    return n;
}

void *ptrt_device_buffer(ptrt_ctx *c, int kind) {
    if (!ctx_live(c))
        return nullptr;
    c->escaped = true; // (the caller may read the context's buffers on streams of its own: frames no longer overlap)
    switch (kind) {
    case PTRT_BUF_ACCUM: return c->d_accum;
    case PTRT_BUF_NORMAL: return c->scaled() ? c->s_normal : c->d_normal;
    case PTRT_BUF_DEPTH: return c->scaled() ? c->s_depth : c->d_depth;
    case PTRT_BUF_OBJECT_ID: return c->scaled() ? c->s_object_id : c->d_object_id;
    case PTRT_BUF_RENDER_ACCUM: return c->scaled() ? c->s_accum : c->d_accum;
    case PTRT_BUF_RGB8: return c->last_rgb8;
    case PTRT_BUF_DENOISED: return c->dn_on ? c->dn_out : nullptr;
    case PTRT_BUF_MOTION: return c->dn_on ? c->dn_motion : nullptr;
    default: return nullptr;
    }
}

int ptrt_read_buffer(ptrt_ctx *c, int kind, void *dst, size_t bytes) {
    if (!ctx_live(c) || !dst)
        return fail(c, PTRT_E_INVALID, "ptrt_read_buffer: bad argument");
    if (int rc = set_device(c))
        return rc;
    size_t need = 0;
    const void *src = nullptr;
    switch (kind) {
    // G-buffers, denoiser images and RENDER_ACCUM hold render-size frames (== the frame size unless
    // ptrt_set_render_size reduced it); ACCUM is the full-size HDR image that was tonemapped
    case PTRT_BUF_ACCUM: need = c->npix * 12; src = c->d_accum; break;
    case PTRT_BUF_RENDER_ACCUM: need = c->rpix() * 12; src = c->scaled() ? c->s_accum : c->d_accum; break;
    case PTRT_BUF_NORMAL: need = c->rpix() * 12; src = c->scaled() ? c->s_normal : c->d_normal; break;
    case PTRT_BUF_DEPTH: need = c->rpix() * 4; src = c->scaled() ? c->s_depth : c->d_depth; break;
    case PTRT_BUF_OBJECT_ID: need = c->rpix() * 4; src = c->scaled() ? c->s_object_id : c->d_object_id; break;
    case PTRT_BUF_RGB8: need = c->npix * 3; src = c->last_rgb8; break;
    case PTRT_BUF_RNG: need = c->npix * 24; break;
    case PTRT_BUF_DENOISED:
    case PTRT_BUF_MOTION:
        if (!c->dn_on)
            return fail(c, PTRT_E_NOT_READY, "ptrt_read_buffer: the denoiser is not enabled");
        need = c->rpix() * (kind == PTRT_BUF_DENOISED ? 12 : 8);
        src = kind == PTRT_BUF_DENOISED ? c->dn_out : c->dn_motion;
        break;
    default: return fail(c, PTRT_E_INVALID, "ptrt_read_buffer: unknown kind %d", kind);
    }
    if (bytes < need)
        return fail(c, PTRT_E_INVALID, "ptrt_read_buffer: destination holds %zu bytes, need %zu", bytes, need);
    if (kind == PTRT_BUF_RGB8 && !src) // (the last frame went straight into a caller's frame: PTRT_OUT_DEVICE_FRAME)
        return fail(c, PTRT_E_NOT_READY, "ptrt_read_buffer: the last frame was written into the caller's frame, the context holds no RGB8 image of it");
    if (kind == PTRT_BUF_RNG) {
        uint32_t *tmp = nullptr;
        HIP_TRY(c, hipMalloc((void **)&tmp, need));
        hipLaunchKernelGGL(pt::rng_planar_to_aos, dim3((unsigned)((c->npix + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_rng, tmp, c->npix);
        hipError_t e = hipMemcpyAsync(dst, tmp, need, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(c->stream);
        (void)hipFree(tmp);
        if (e != hipSuccess)
            return fail(c, PTRT_E_HIP, "RNG read-back failed: %s", hipGetErrorString(e));
        return PTRT_OK;
    }
    HIP_TRY(c, hipMemcpyAsync(dst, src, need, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PTRT_OK;
}

int ptrt_write_rng(ptrt_ctx *c, const uint32_t *states, size_t bytes) {
    if (!ctx_live(c) || !states || bytes < c->npix * 24)
        return fail(c, PTRT_E_INVALID, "ptrt_write_rng: bad argument");
    if (int rc = set_device(c))
        return rc;
    uint32_t *tmp = nullptr;
    HIP_TRY(c, hipMalloc((void **)&tmp, c->npix * 24));
    hipError_t e = hipMemcpyAsync(tmp, states, c->npix * 24, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pt::rng_aos_to_planar, dim3((unsigned)((c->npix + 255) / 256)), dim3(256), 0, c->stream, tmp,
                           c->d_rng, c->npix);
        e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(tmp);
    if (e != hipSuccess)
        return fail(c, PTRT_E_HIP, "RNG upload failed: %s", hipGetErrorString(e));
    c->rng_ready = true;
    return PTRT_OK;
}

int ptrt_trace_rays(ptrt_ctx *c, const float *origins, const float *directions, int n, ptrt_hit *out) {
    static_assert(sizeof(pt::HitOut) == sizeof(ptrt_hit), "HitOut must mirror ptrt_hit");
    if (!ctx_live(c) || !origins || !directions || !out || n < 0)
        return fail(c, PTRT_E_INVALID, "ptrt_trace_rays: bad argument");
    if (!c->have_geometry)
        return fail(c, PTRT_E_NOT_READY, "ptrt_trace_rays: geometry not uploaded");
    if (n == 0)
        return PTRT_OK;
    if (int rc = set_device(c))
        return rc;
    float *d_o = nullptr, *d_d = nullptr;
    pt::HitOut *d_h = nullptr;
    int rc = PTRT_OK;
    hipError_t e = hipMalloc((void **)&d_o, (size_t)n * 12);
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_d, (size_t)n * 12);
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_h, (size_t)n * sizeof(pt::HitOut));
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_o, origins, (size_t)n * 12, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_d, directions, (size_t)n * 12, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        pt::KParams K = make_params(c);
        const int geom = pick_geom(c);
        const size_t lds = (geom == 0) ? 0 : (size_t)c->stack_entries * 64 * sizeof(uint2);
        const int grid = (n + 63) / 64;
        if (geom == 0)
            hipLaunchKernelGGL(pt::trace_rays_kernel<0>, dim3(grid), dim3(64), lds, c->stream, K, d_o, d_d, n, d_h);
        else if (geom == 1)
            hipLaunchKernelGGL(pt::trace_rays_kernel<1>, dim3(grid), dim3(64), lds, c->stream, K, d_o, d_d, n, d_h);
        else
            hipLaunchKernelGGL(pt::trace_rays_kernel<2>, dim3(grid), dim3(64), lds, c->stream, K, d_o, d_d, n, d_h);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(out, d_h, (size_t)n * sizeof(pt::HitOut), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess)
        rc = fail(c, PTRT_E_HIP, "ptrt_trace_rays: %s", hipGetErrorString(e));
    (void)hipFree(d_o);
    (void)hipFree(d_d);
    (void)hipFree(d_h);
    return rc;
}

int ptrt_get_stats(ptrt_ctx *c, ptrt_stats *out) {
    if (!ctx_live(c) || !out)
        return fail(c, PTRT_E_INVALID, "ptrt_get_stats: bad argument");
    if (int rc = set_device(c))
        return rc;
    // per-workgroup slots (no atomics in the kernel, see pt_render.hip.h): summed here
    std::vector<unsigned long long> slots(c->n_counter_slots * pt::COUNTER_WORDS);
    HIP_TRY(c, hipMemcpyAsync(slots.data(), c->d_counters, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                              c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, slots.size() * sizeof(unsigned long long), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long h[pt::COUNTER_WORDS] = {0, 0, 0, 0};
    for (size_t i = 0; i < slots.size(); ++i)
        h[i % pt::COUNTER_WORDS] += slots[i];
    out->extension_rays = h[0];
    out->shadow_rays = h[1];
    out->paths = h[2];
    out->shadow_rays_walked = h[1] - h[3];
    return PTRT_OK;
}

int ptrt_set_option(ptrt_ctx *c, const char *name, long long value) {
    if (!ctx_live(c, false) || !name)
        return fail(c, PTRT_E_INVALID, "ptrt_set_option: bad argument");
    const std::string n(name);
    if (n == "count_rays")
        c->count_rays = value ? 1 : 0;
    else if (n == "force_geom") { // -1 auto; 1 / 2 force a more general traversal variant (tests)
        if (value < -1 || value > 2)
            return fail(c, PTRT_E_INVALID, "force_geom must be -1..2");
        c->force_geom = (int)value;
    } else if (n == "force_full")
        c->force_full = value ? 1 : 0;
    else if (n == "pair_trace") // 0: lock-step mesh loop instead of (ray, mesh) pair compaction (A/B, tests)
        c->pair_trace = value ? 1 : 0;
    else if (n == "steal") { // PMODE 2 shadow rays: 0 = no subtree stealing; n = node steps between steal rounds
        if (value < 0 || value > 64)
            return fail(c, PTRT_E_INVALID, "steal must be 0..64");
        c->steal = (int)value;
    } else if (n == "csteal") { // PMODE 2 closest hit: 0 = no subtree stealing; n = node steps between steal rounds (verified: same bits)
        if (value < 0 || value > 64)
            return fail(c, PTRT_E_INVALID, "csteal must be 0..64");
        c->csteal = (int)value;
    } else if (n == "atrous_exp")
        c->atrous_exp = value ? 1 : 0;
    else if (n == "csteal_follow")
        c->csteal_follow = value ? 1 : 0;
    else if (n == "csteal_leaf_min") {
        if (value < 1 || value > 64)
            return fail(c, PTRT_E_INVALID, "csteal_leaf_min must be 1..64");
        c->csteal_leaf_min = (int)value;
    }
    else if (n == "csteal_min") {
        if (value < 0 || value > 1024)
            return fail(c, PTRT_E_INVALID, "csteal_min must be 0..1024");
        c->csteal_min = (int)value;
    } else if (n == "lds_nodes") // PMODE 2 in 4-wave workgroups with the BLAS top levels staged in LDS (A/B, tests)
        c->lds_nodes = value < 0 ? 0 : (value > 2 ? 2 : (int)value); // (2: the larger workgroups without reading the staged nodes)
    else if (n == "merged") // PMODE 4 instead of 2: shadow rays ride with the next extension rays (A/B, tests)
        c->merged = value < 0 ? -1 : (value ? 1 : 0);
    else if (n == "leaf_pairs") // PMODE 2: 0 = every lane walks its own leaf (A/B, tests)
        c->leaf_pairs = value ? 1 : 0;
    else if (n == "lds_pad") { // extra bytes of LDS per workgroup: fewer waves per CU (A/B of the occupancy, tests)
        if (value < 0 || value > 32768)
            return fail(c, PTRT_E_INVALID, "lds_pad must be 0..32768");
        c->lds_pad = (int)value;
    }
    else if (n == "time_kernels") // 0: no start / stop events around the trace kernel (two driver calls per frame; ptrt_kernel_ms_history then has nothing)
        c->time_kernels = value ? 1 : 0;
    else if (n == "time_launches") // 1: events around every launch of a frame dealt to the auxiliary streams (ptrt_launch_ms_history)
        c->time_launches = value ? 1 : 0;
    else if (n == "tm_prio") // lane refill's tonemap pass: | 1 on a stream of the highest priority, | 2 its waves at s_setprio 3
        c->tm_prio = (int)(value & 3);
    else if (n == "pipeline") // 1 (default): consecutive frames may overlap on the device when that is safe (ptrt_render); 0: never
        c->pipeline = value ? 1 : 0;
    else if (n == "persist")
        c->persist = value < 0 ? 0 : value;
    else if (n == "sample_sync")
        c->sample_sync = value < 0 ? -1 : (value != 0);
    else if (n == "tile_run") {
        if (value < 0 || value > 64)
            return fail(c, PTRT_E_INVALID, "tile_run: 0 (tile k on workgroup k) or the tiles per XCD and run, 1..64");
        c->tile_run = value;
    }
    else if (n == "ticket_tiles")
        c->ticket_tiles = value < 1 ? 1 : (value > 16 ? 16 : (int)value);
    else if (n == "refill")
        c->refill = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (n == "split") { // tile rows of the frame dealt to that many concurrent launches of the megakernel (1 = one launch)
        if (value < 1 || value > ptrt_ctx::MAX_SPLIT)
            return fail(c, PTRT_E_INVALID, "split must be 1..%d", ptrt_ctx::MAX_SPLIT);
        c->split = (int)value;
    } else if (n == "tlas_rounds") // PMODE 3 shadow rays: one TLAS leaf per ray and fill instead of all of them (A/B, tests)
        c->tlas_rounds = value ? 1 : 0;
    else if (n == "pm1_wg") { // PMODE 1: one or two tiles per workgroup (0 = choose by the LDS budget; A/B, tests)
        if (value < 0 || value > 20)
            return fail(c, PTRT_E_INVALID, "pm1_wg must be 0..2");
        c->pm1_wg = (int)value;
    }
    else if (n == "stage") // PMODE 1: shading inputs staged in LDS (0 none; else jitter inputs, | 1 lights, | 2 materials; A/B, tests)
        c->stage = (int)(value & 7);
    else if (n == "pair_split") // PMODE 1: 0 = one lane per pair also in batches that do not fill the wave (A/B, tests)
        c->pair_split = value ? 1 : 0;
    else if (n == "async_lanes") // 1: persistent megakernel with asynchronous lanes for single-leaf-TLAS scenes
        c->async_lanes = value ? 1 : 0;
    else if (n == "shade_min") { // async_lanes: lanes that wait for the shading block before it runs
        if (value < 1 || value > 64)
            return fail(c, PTRT_E_INVALID, "shade_min must be 1..64");
        c->shade_min = (int)value;
    } else if (n == "leaf_min") { // PMODE 2 and async_lanes: lanes waiting at a leaf that end the node loop
        if (value < 1 || value > 64)
            return fail(c, PTRT_E_INVALID, "leaf_min must be 1..64");
        c->leaf_min = c->as_leaf_min = (int)value;
    } else if (n == "wavefront") // 1: trace/shade stages over the whole frame's rays instead of the megakernel
        c->wavefront = value ? 1 : 0;
    else if (n == "wf_sort") // wavefront stages: the shade stage sorts its paths by class (material, bounce) in LDS first, 1 / 2 / 4 groups of 256 together
        c->wf_sort = value <= 0 ? 0 : (value >= 4 ? 4 : (value >= 2 ? 2 : 1));
    else if (n == "fetch_min") { // PMODE 2: refill threshold in idle lanes; 0 = static batches of 64 pairs (A/B, tests)
        if (value < 0 || value > 64)
            return fail(c, PTRT_E_INVALID, "fetch_min must be 0..64");
        c->fetch_min = (int)value;
    } else if (n == "denoiser_active") // perfSettings.enableDenoiser: use the (already allocated) denoiser or not
        c->dn_active = value ? 1 : 0;
    else if (n == "motion_vectors") // perfSettings.enableMotionVectors
        c->mv_active = value ? 1 : 0;
    else if (n == "use_graphs") // 0: issue the refit / rebuild launches one by one instead of replaying a hipGraph
        c->use_graphs = value ? 1 : 0;
    else
        return fail(c, PTRT_E_INVALID, "unknown option '%s'", name);
    return PTRT_OK;
}

// what ptrt_set_option set, plus read-only facts about the last ptrt_render (so that a measurement can say what ran)
int ptrt_get_option(ptrt_ctx *c, const char *name, long long *value) {
    if (!ctx_live(c, false) || !name || !value)
        return fail(c, PTRT_E_INVALID, "ptrt_get_option: bad argument");
    const std::string n(name);
    const std::pair<const char *, long long> tab[] = {
        {"count_rays", c->count_rays}, {"force_geom", c->force_geom}, {"force_full", c->force_full}, {"pair_trace", c->pair_trace},
        {"steal", c->steal}, {"csteal", c->csteal}, {"csteal_min", c->csteal_min}, {"csteal_follow", c->csteal_follow}, {"atrous_exp", c->atrous_exp}, {"csteal_leaf_min", c->csteal_leaf_min}, {"lds_nodes", c->lds_nodes}, {"merged", c->merged}, {"leaf_pairs", c->leaf_pairs}, {"lds_pad", c->lds_pad},
        {"stage", c->stage}, {"pm1_wg", c->pm1_wg}, {"tlas_rounds", c->tlas_rounds}, {"time_kernels", c->time_kernels}, {"time_launches", c->time_launches}, {"tm_prio", c->tm_prio}, {"persist", c->persist}, {"refill", c->refill}, {"sample_sync", c->sample_sync}, {"sample_sync_eff", c->sample_sync_eff}, {"tile_run", c->tile_run}, {"ticket_tiles", c->ticket_tiles}, {"refilled", c->refill_eff ? 1 : 0}, {"split", c->split}, {"split_eff", c->split_eff}, {"pipeline", c->pipeline}, {"pipelined", c->pipelined_last ? 1 : 0}, {"pair_split", c->pair_split}, {"async_lanes", c->async_lanes}, {"shade_min", c->shade_min},
        {"leaf_min", c->leaf_min}, {"wavefront", c->wavefront}, {"wf_sort", c->wf_sort}, {"fetch_min", c->fetch_min}, {"denoiser_active", c->dn_active},
        {"motion_vectors", c->mv_active}, {"use_graphs", c->use_graphs},
        // read-only: the last launch
        {"render_mode", c->last_mode},   // 0 megakernel, 1 wavefront stages, 2 asynchronous lanes
        {"pmode", c->last_pmode},        // PMODE of the megakernel: 0 lock-step, 1 pairs/LDS triangles, 2 queue, 3 TLAS rounds, 4 merged queue
        {"merged_eff", c->merged_eff},   // loop shape of the last launch (1 = shadow rays ride with the next extension rays)
        {"merged_decided", (c->merged >= 0 || c->tune_choice >= 0 || !c->last_merged_possible) ? 1 : 0}, // 0 while "merged" = -1 is still sampling
        {"launches", (long long)c->launches},
    };
    for (const auto &e : tab)
        if (n == e.first) {
            *value = e.second;
            return PTRT_OK;
        }
    return fail(c, PTRT_E_INVALID, "unknown option '%s'", name);
}

// test hook: which kernels rendered the last frame (0 megakernel, 1 wavefront stages)
int ptrt_debug_last_render_mode(ptrt_ctx *c) { return ctx_live(c) ? c->last_mode : -1; }

// profiling hook (not part of the drop-in surface): reads and clears pt::g_trav_stats (32 words); all zero unless
// the library was built with -DPT_TRAV_STATS
int ptrt_debug_trav_stats(ptrt_ctx *c, unsigned long long *out32) {
    if (!ctx_live(c) || !out32)
        return fail(c, PTRT_E_INVALID, "ptrt_debug_trav_stats: bad argument");
    if (int rc = set_device(c))
        return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out32, HIP_SYMBOL(pt::g_trav_stats), 32 * sizeof(unsigned long long)));
    unsigned long long zero[32] = {};
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(pt::g_trav_stats), zero, sizeof(zero)));
    return PTRT_OK;
}

// ... and pt::g_trav_bounce (64 words: the sixteen traversal counters split by the rays' bounce 0, 1, 2, >= 3)
int ptrt_debug_trav_bounce(ptrt_ctx *c, unsigned long long *out64) {
    if (!ctx_live(c) || !out64)
        return fail(c, PTRT_E_INVALID, "ptrt_debug_trav_bounce: bad argument");
    if (int rc = set_device(c))
        return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out64, HIP_SYMBOL(pt::g_trav_bounce), 64 * sizeof(unsigned long long)));
    unsigned long long zero[64] = {};
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(pt::g_trav_bounce), zero, sizeof(zero)));
    return PTRT_OK;
}

#ifdef PT_TRAV_STATS
int ptrt_debug_trav_dbg(ptrt_ctx *c, unsigned long long *out1033) {
    if (!ctx_live(c) || !out1033)
        return fail(c, PTRT_E_INVALID, "ptrt_debug_trav_dbg: bad argument");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out1033, HIP_SYMBOL(pt::g_trav_dbg), 1033 * sizeof(unsigned long long)));
    static unsigned long long zero[1033] = {};
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(pt::g_trav_dbg), zero, sizeof(zero)));
    return PTRT_OK;
}
#endif

// test hook: exhaustive rcp_ieee check; out9[0] = mismatches, out9[1..8] = first offending inputs
int ptrt_debug_rcp_check(ptrt_ctx *c, unsigned int *out9) {
    if (!ctx_live(c) || !out9)
        return fail(c, PTRT_E_INVALID, "ptrt_debug_rcp_check: bad argument");
    if (int rc = set_device(c))
        return rc;
    unsigned int *d = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d, 9 * sizeof(unsigned int)));
    HIP_TRY(c, hipMemset(d, 0, 9 * sizeof(unsigned int)));
    hipLaunchKernelGGL(pt::rcp_check_kernel, dim3(4096), dim3(256), 0, c->stream, d);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out9, d, 9 * sizeof(unsigned int), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return PTRT_OK;
}

// test hook: exhaustive sqrt_ieee check; out9[0] = mismatches, out9[1] = mismatches of the bare core in its range, out9[2..8] = inputs
int ptrt_debug_sqrt_check(ptrt_ctx *c, unsigned int *out9) {
    if (!ctx_live(c) || !out9)
        return fail(c, PTRT_E_INVALID, "ptrt_debug_sqrt_check: bad argument");
    if (int rc = set_device(c))
        return rc;
    unsigned int *d = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d, 9 * sizeof(unsigned int)));
    HIP_TRY(c, hipMemset(d, 0, 9 * sizeof(unsigned int)));
    hipLaunchKernelGGL(pt::sqrt_check_kernel, dim3(4096), dim3(256), 0, c->stream, d);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out9, d, 9 * sizeof(unsigned int), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return PTRT_OK;
}

// test hook: div3's core for the divisors 1.m, m in [first, first + count), against every numerator significand (mode 0), or
// div3 with out-of-range exponents (modes 1, 2); out9[0] = mismatches, out9[1..8] = first offending {a, t} bit patterns
int ptrt_debug_div3_check(ptrt_ctx *c, unsigned int first, unsigned int count, int mode, unsigned int *out9) {
    if (!ctx_live(c) || !out9 || count == 0 || count > (1u << 23) || mode < 0 || mode > 3)
        return fail(c, PTRT_E_INVALID, "ptrt_debug_div3_check: bad argument");
    if (int rc = set_device(c))
        return rc;
    unsigned int *d = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d, 9 * sizeof(unsigned int)));
    HIP_TRY(c, hipMemset(d, 0, 9 * sizeof(unsigned int)));
    hipLaunchKernelGGL(pt::div3_check_kernel, dim3((count + 63) / 64), dim3(64), 0, c->stream, first, count, mode, d);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out9, d, 9 * sizeof(unsigned int), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return PTRT_OK;
}

// test hook (not part of the drop-in surface): the kernels' deterministic math on the GPU
int ptrt_debug_detmath(ptrt_ctx *c, int op, const float *x, const float *y, int n, float *out) {
    if (!ctx_live(c) || !x || !out || n <= 0)
        return fail(c, PTRT_E_INVALID, "ptrt_debug_detmath: bad argument");
    if (int rc = set_device(c))
        return rc;
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    HIP_TRY(c, hipMalloc((void **)&dx, (size_t)n * 4));
    HIP_TRY(c, hipMalloc((void **)&dy, (size_t)n * 4));
    HIP_TRY(c, hipMalloc((void **)&dout, (size_t)n * 4));
    HIP_TRY(c, hipMemcpy(dx, x, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(dy, y ? y : x, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pt::detmath_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, op, dx, dy, n, dout);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx);
    (void)hipFree(dy);
    (void)hipFree(dout);
    return PTRT_OK;
}

// test hook: the shading functions one by one on the device (pt::shade_probe_kernel) over the context's uploaded materials;
// op 0: n items of 11 floats -> 4 floats each; op 1: n items of 14 -> 13 (see the kernel).  full = 0: the simple-material variant
int ptrt_debug_shade(ptrt_ctx *c, int op, int full, const float *in, int n, float *out) {
    if (!ctx_live(c) || !in || !out || n <= 0 || (op != 0 && op != 1))
        return fail(c, PTRT_E_INVALID, "ptrt_debug_shade: bad argument");
    if (!c->have_materials)
        return fail(c, PTRT_E_NOT_READY, "ptrt_debug_shade: materials not uploaded");
    if (int rc = set_device(c))
        return rc;
    const size_t ni = (size_t)n * (op == 0 ? 11 : 14) * 4, no = (size_t)n * (op == 0 ? 4 : 13) * 4;
    float *din = nullptr, *dout = nullptr;
    HIP_TRY(c, hipMalloc((void **)&din, ni));
    HIP_TRY(c, hipMalloc((void **)&dout, no));
    HIP_TRY(c, hipMemcpy(din, in, ni, hipMemcpyHostToDevice));
    if (full)
        hipLaunchKernelGGL(pt::shade_probe_kernel<true>, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->d_materials, op, din, n, dout);
    else
        hipLaunchKernelGGL(pt::shade_probe_kernel<false>, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->d_materials, op, din, n, dout);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, dout, no, hipMemcpyDeviceToHost));
    (void)hipFree(din);
    (void)hipFree(dout);
    return PTRT_OK;
}

} // extern "C"

#include "ptrt_farm.hip.h"

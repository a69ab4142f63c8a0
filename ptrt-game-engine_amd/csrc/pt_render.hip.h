// pt_render.hip.h -- path_trace_kernel<GEOM,FULL,PMODE>: the render loop.
//
// Same outputs, bit for bit, as the first cut; what changed is how a wave spends its lanes:
//
//  * wave-uniform phases.  One iteration of the persistent loop is
//        [A] regenerate finished paths            (divergent, cheap)
//        [B] closest hit for every live lane      (wave-uniform call)
//        [C] shade: emission, light sample        (divergent) -> shadow ray per lane
//        [D] shadow any-hit for every lane that has one (wave-uniform call)
//        [E] add the light sample, sample the BSDF, roulette, next ray (divergent)
//    so both traversals run with all live lanes and may use wave collectives.
//
//  * PAIRS (scenes whose BLASes are all single leaves, e.g. the Cornell box): in-wave
//    compaction of (ray, mesh) pairs.  A lane hits the root box of only ~1.7 of the 8
//    Cornell meshes, but a lock-step mesh loop makes every lane sit through the union
//    (~6-8 meshes) -> ~25 % useful lanes in the triangle tests.  Instead:
//      1. every lane slab-tests all M root boxes (scalar-loaded boxes), and for each mesh
//         the hitting lanes append {lane, mesh order} to an LDS pair list at
//         base + mbcnt(ballot)                        (ballot / prefix-sum compaction)
//      2. the P pairs are processed 64 at a time: lane j takes pair j, fetches that ray
//         from LDS, walks the mesh's triangle packets (staged once per workgroup in LDS)
//         and keeps the pair's closest hit
//      3. per-ray merge with one LDS 64-bit atomic min on {t bits, mesh order, leaf index}:
//         smallest t wins, ties go to the earlier mesh, then the earlier triangle -- exactly
//         the reference's strict-`<` first-minimum (equivalence E3 in pt_kernels.hip.h).
//    Shadow rays use the same pairs with an LDS flag per ray instead of the min.
//    A batch that does not fill the wave gives every pair 2^k lanes (pair_split).
//
//  * PMODE 2 (real BLASes behind a single-leaf TLAS) keeps the pair list as a QUEUE (run_closest_queue /
//    run_any_queue): idle lanes refill by ballot rank; the node loop yields once leaf_min lanes wait at a
//    leaf; the leaf phase runs as compacted (lane, triangle) pairs fed by ds_bpermute from the owner lane's
//    registers; shadow rays share subtrees by stealing the bottom of busy lanes' stacks.  Measured lane
//    occupancy before/after: profiles/r01e_lane_occupancy.txt, DESIGN.md 3.1.
//  * PMODE 3 (a real TLAS) runs that queue in rounds, one TLAS leaf per ray per round.
//  * PMODE 0: lock-step fall-backs (and the A/B baseline of the tests).
#pragma once
#include "pt_kernels.hip.h"

namespace pt {

struct CycleAcc;
struct PairLds {
    CycleAcc *cyc;            // (stats build: the kernel's cycle accumulator)
    int stat_bounce;          // (stats build: bounce of the rays of this traversal -- wave-uniform with the samples in step)
    float4 *tris;             // pair_tri_slots * 3
    int4 *meshtab;            // per mesh order: {first slot, count, flags, mesh id}
    float4 *meshbox;          // per mesh order: {bmin, root ref}, {bmax, flags}: the mesh record's head, staged once
    const float4 *topnodes;   // WG = 4 variant: the first TOP_NODES nodes of every mesh of the leaf, [order][node][4], or NULL
    uint32_t *pairs;          // 64 * pair_meshes
    unsigned long long *best; // 64
    uint32_t *occ;            // 64
    uint2 *stack;             // PMODE 2: [entry][lane] BLAS traversal stack
    // PMODE 2, (lane, triangle) pair compaction of the leaf phase (leaf_pairs)
    int *leafx;               // PMODE 3: 64: first TLAS index of the leaf each ray is at
    uint2 *tstack;            // PMODE 3: [entry][lane] TLAS traversal stack
    unsigned long long *lkey; // 64: {t bits, index in leaf} min per lane
    unsigned char *owner;     // 64 * 17 rounded up: lane of each test
    // shading inputs staged behind the lists (KParams::lds_extra; one-wave workgroups of PMODE >= 1), else NULL
    const float2 *jit;        // 16: TAA jitter table
    const float2 *bn;         // 64: the lanes' blue-noise values
    const float4 *lights;     // the scene's light records (4 float4 each) when at most LDS_LIGHTS of them
    const float4 *mats;       // PMODE 1: material records (6 float4) of the leaf's meshes by mesh order
    unsigned long long *dirty; // PMODE 2, closest-hit stealing: bit r = ray r has to be traced again without it (run_closest_queue); or NULL
    unsigned long long *count; // one-wave workgroups: [0] {extension rays, shadow rays << 32} of the wave so far, [1] light
                               // samples whose value was exactly zero (counted in [0], not walked)
};
constexpr int LDS_LIGHTS = 8;
constexpr int LDS_EXTRA_FIXED = 16 * 8 + 64 * 8; // jitter table + blue-noise values
constexpr int LEAF_PAIR_BYTES = 512 + 1088; // lkey + owner (64 lanes x 17 tests)
// PMODE 3: TLAS leaves a ray may have in one fill of the pair list (a power of two <= 4) and the pairs that end a fill
#ifndef PT_TLAS_SLOTS
#define PT_TLAS_SLOTS 1
#endif
constexpr int TLAS_SLOTS = PT_TLAS_SLOTS;
constexpr int TLAS_FILL_TARGET = 64;
constexpr int PAIR_PAD = 2; // float4 of padding in front of each mesh's packets in LDS (bank spreading)
PT_DEV PairLds carve_pair_lds(void *base, int tri_slots, int meshes, int stack_entries = 0, int tlas_leaf = 0,
                              int tlas_depth = 0, int pair_cap = 0) {
    PairLds l{};
    char *p = (char *)base;
    l.tris = (float4 *)p;
    p += tri_slots ? ((size_t)tri_slots * 48 + (size_t)meshes * PAIR_PAD * 16) : 0;
    l.meshtab = (int4 *)p; // (PMODE 1 only: {first slot, count, flags, mesh id}; the other modes read the staged heads)
    p += tri_slots ? (size_t)meshes * 16 : 0;
    l.meshbox = (float4 *)p;
    p += (size_t)meshes * 32;
    l.best = (unsigned long long *)p;
    // the any-hit traversals use the first half of `best` as a plane of ray limits; their blocked flags take the second
    // (PMODE 4 walks both kinds of ray at once and keeps its flags apart)
    l.occ = (uint32_t *)(p + 256);
    p += tlas_leaf ? 512 * TLAS_SLOTS : 512; // (PMODE 3: one minimum per ray and leaf slot)
    l.pairs = (uint32_t *)p;
    // 16-bit entries {lane, mesh order << 6}; PMODE 3 holds one TLAS leaf per ray at a time
    // (PMODE 4 keeps the pairs of both ray kinds in one list of pair_cap 16-bit entries)
    p += pair_cap ? (size_t)pair_cap * 2 : tlas_leaf ? ((size_t)tlas_leaf * 64 + TLAS_FILL_TARGET) * 2 : (size_t)meshes * 128;
    if (pair_cap) {
        l.occ = (uint32_t *)p;
        p += 256;
    }
    l.stack = (uint2 *)p;
    p += (size_t)stack_entries * 512;
    l.tstack = (uint2 *)p;
    p += (size_t)tlas_depth * 512;
    l.leafx = (int *)p;
    p += tlas_leaf ? 256 * TLAS_SLOTS : 0;
    l.lkey = (unsigned long long *)p;
    p += 512;
    l.owner = (unsigned char *)p;
    p += LEAF_PAIR_BYTES - 512;
    // the wave's ray totals (path_trace_kernel): PMODE 1 keeps them in the spare bytes behind its last mesh's packets
    l.count = tri_slots ? (unsigned long long *)(l.tris + tri_slots * 3 + (meshes - 1) * PAIR_PAD) : (unsigned long long *)p;
    l.dirty = tri_slots ? nullptr : (unsigned long long *)p + 2;
    return l;
}

// PMODE 1 (every BLAS one leaf, the scene's triangles in LDS).  A workgroup of WG waves = WG neighbouring tiles shares ONE copy
// of what is read-only -- triangle packets, mesh table, mesh heads, jitter table, light and material records -- and each
// wave has its own lists behind it: the rays' minima / flags, the pair list, the lanes' blue-noise values, the ray totals.
// Layout (host: pm1_layout in ptrt_capi.hip): [tris + pads | meshtab | meshbox] [lds_extra: jit | lights | mats]
// [lds_wave + wave * lds_wave_bytes: best 512 | pairs 128 * meshes | bn 512 (if staged) | count 16].
// PMODE 1, per wave: the minima (512 B) and the pair list behind them are scratch of the traversal phases [B] / [D]; the
// lane-refill kernel parks a finished pixel's six generator words there across [R] (6 x 256 B), so the list is never
// smaller than what is left of that after the minima
__host__ __device__ constexpr size_t pm1_pair_bytes(int meshes) { return (size_t)meshes * 128 > 1024 ? (size_t)meshes * 128 : 1024; }
PT_DEV PairLds carve_pm1(void *base, const KParams &K, int wave) {
    PairLds l{};
    char *p = (char *)base;
    const int meshes = K.pair_meshes;
    l.tris = (float4 *)p;
    p += (size_t)K.pair_tri_slots * 48 + (size_t)meshes * PAIR_PAD * 16;
    l.meshtab = (int4 *)p;
    p += (size_t)meshes * 16;
    l.meshbox = (float4 *)p;
    char *x = (char *)base + K.lds_extra;
    l.jit = (const float2 *)x;
    l.lights = (const float4 *)(x + 16 * 8);
    l.mats = l.lights + ((K.lds_flags & 1) ? K.n_lights * 4 : 0);
    char *w = (char *)base + K.lds_wave + wave * K.lds_wave_bytes;
    l.best = (unsigned long long *)w;
    l.occ = (uint32_t *)(w + 256); // (the any-hit flags take the second half of the minima's words)
    l.pairs = (uint32_t *)(w + 512);
    w += 512 + pm1_pair_bytes(meshes);
    l.bn = (const float2 *)w;
    w += (K.lds_flags & 8) ? 64 * 8 : 0;
    l.count = (unsigned long long *)w;
    return l;
}

// PMODE 2 in 256-thread workgroups (path_trace_kernel<.., WG = 4>): mesh table, mesh heads and the top levels of every
// BLAS are ONE copy per workgroup at the base of its LDS; behind them each wave has its own lists and stacks.
PT_DEV size_t shared_lds_bytes(int meshes) { return (size_t)meshes * (32 + TOP_NODES * 64); }
PT_DEV size_t wave_lds_bytes(int meshes, int stack_entries) {
    return 512 + (size_t)meshes * 128 + 256 + (size_t)stack_entries * 512 + LEAF_PAIR_BYTES;
}
PT_DEV PairLds carve_pair_lds_wg(void *base, int wave, int meshes, int stack_entries) {
    PairLds l{};
    char *p = (char *)base;
    l.meshtab = nullptr;
    l.meshbox = (float4 *)p;
    p += (size_t)meshes * 32;
    l.topnodes = (const float4 *)p;
    p += (size_t)meshes * TOP_NODES * 64;
    p += (size_t)wave * wave_lds_bytes(meshes, stack_entries);
    l.best = (unsigned long long *)p;
    p += 512;
    l.pairs = (uint32_t *)p;
    p += (size_t)meshes * 128;
    l.occ = (uint32_t *)p;
    p += 256;
    l.stack = (uint2 *)p;
    p += (size_t)stack_entries * 512;
    l.lkey = (unsigned long long *)p;
    p += 512;
    l.owner = (unsigned char *)p;
    l.tris = nullptr;
    l.tstack = nullptr;
    l.leafx = nullptr;
    return l;
}

// Traversal statistics (build with -DPT_TRAV_STATS; read with ptrt_debug_trav_stats): how full the
// wave is in each loop of the PMODE 2 traversals.  [0..7] closest, [8..15] any-hit:
// calls, pairs, node wave-iterations, node lane-steps, leaf phases, triangle wave-iterations,
// triangle lane-tests, outer iterations; [16] persistent-loop iterations, [17] live lanes in them.
__device__ unsigned long long g_trav_bounce[64]; // the traversal counters [0..15] once more, split by the rays' bounce: [bounce 0..3][16]
__device__ unsigned long long g_trav_stats[32]; // [0..7] closest, [8..15] any-hit, [16..17] loop, [18..23] lanes per phase (TS_LANES), [24..31] cycles (CycleAcc)
// -DPT_MARKS: "; MARK x" comments in the ISA at the phase boundaries of path_trace_kernel (tools/asm_phases.py counts the
// instructions between them)
#ifdef PT_MARKS
#define PT_MARK(x) asm volatile("; MARK " x)
#else
#define PT_MARK(x)
#endif
#ifdef PT_TRAV_STATS
struct TravStats {
    unsigned v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PT_DEV void wave(int i, int lane) { // once per wave-level execution of the enclosing block
        if (lane == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
            ++v[i];
    }
    PT_DEV void lanes(int i) { ++v[i]; }
    PT_DEV void flush(int base, int lane, int bounce = -1) {
        for (int i = 0; i < 8; ++i) {
            unsigned a = v[i];
            for (int off = 32; off > 0; off >>= 1)
                a += __shfl_xor(a, off);
            if (lane == 0 && a) {
                atomicAdd(&g_trav_stats[base + i], (unsigned long long)a);
                if (bounce >= 0)
                    atomicAdd(&g_trav_bounce[(bounce < 3 ? bounce : 3) * 16 + base + i], (unsigned long long)a);
            }
        }
    }
};
__device__ unsigned long long g_trav_dbg[4 * 256 + 1 + 8]; // 256 records, their count, event counters [1025 + k]
#define TS_DBG(a, b, c, d)                                                                                                \
    do {                                                                                                                 \
        const unsigned long long i_ = atomicAdd(&g_trav_dbg[1024], 1ull);                                                \
        if (i_ < 256) {                                                                                                  \
            g_trav_dbg[4 * i_] = (a);                                                                                    \
            g_trav_dbg[4 * i_ + 1] = (b);                                                                                \
            g_trav_dbg[4 * i_ + 2] = (c);                                                                                \
            g_trav_dbg[4 * i_ + 3] = (d);                                                                                \
        }                                                                                                                \
    } while (0)
#define TS_EVENT(slot) atomicAdd(&g_trav_dbg[1025 + (slot)], 1ull) // (events: 0 subtrees stolen, 1 thief hits in front of their leaf box, 2 equal distances of two walks of a pair, 3 rays traced again)
#define TS_WAVE(i) ts.wave(i, lane)
#define TS_LANE(i) ts.lanes(i)
// wave-level cycle accounting (s_memtime), accumulated in scalar registers and flushed once per wave at kernel end
// (an atomic per interval would be what the kernel spends its time on)
struct CycleAcc {
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PT_DEV void flush(int lane) {
        if (lane == 0)
            for (int i = 0; i < 8; ++i)
                atomicAdd(&g_trav_stats[24 + i], c[i]);
    }
};
// lanes for which `cond` holds, summed over the wave-level executions of the statement ([18..23]: per-phase participation)
#define TS_LANES(slot, cond)                                                                                             \
    do {                                                                                                                 \
        const unsigned long long ts_m = __builtin_amdgcn_ballot_w64(cond);                                               \
        if (lane == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))                                                   \
            atomicAdd(&g_trav_stats[slot], (unsigned long long)__builtin_popcountll(ts_m));                                 \
    } while (0)
#define TS_NOW() __builtin_readcyclecounter()
#define TS_ADD(slot, t) (cyc.c[(slot) - 8] += __builtin_readcyclecounter() - (t))
#define TS_ADDL(slot, t) (L.cyc->c[(slot) - 8] += __builtin_readcyclecounter() - (t))
// slots 8..11 hold the traversal queues' parts (queue run, node loops, leaf blocks, any-hit run) -- or, in a build with
// -DPT_TS_SHADING as well, the shading phases [A] / [C] / [C2] / [E] of every loop shape (PMODE 1, which has no queues, always)
#ifdef PT_TS_SHADING
constexpr bool TS_SHADING = true;
#else
constexpr bool TS_SHADING = false;
#endif
#define TS_ADDQ(slot, t) do { if (!TS_SHADING) TS_ADD(slot, t); } while (0)
#define TS_ADDLQ(slot, t) do { if (!TS_SHADING) TS_ADDL(slot, t); } while (0)
#else
constexpr bool TS_SHADING = false;
#define TS_ADDQ(slot, t) (void)(t)
#define TS_ADDLQ(slot, t) (void)(t)
struct TravStats {
    PT_DEV void flush(int, int, int = -1) {}
};
#define TS_EVENT(slot)
#define TS_DBG(a, b, c, d)
#define TS_WAVE(i)
#define TS_LANE(i)
struct CycleAcc {
    PT_DEV void flush(int) {}
};
#define TS_LANES(slot, cond)
#define TS_NOW() 0ull
#define TS_ADD(slot, t) (void)(t)
#define TS_ADDL(slot, t) (void)(t)
#endif

PT_DEV void wave_lds_order() { // LDS is in order within a wave; this only pins the compiler
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// The pair lists, merge keys and stacks of a trace belong to ONE wave (a 64-thread workgroup, or one wave's slice of
// a larger workgroup's LDS): between its phases the wave only needs its own LDS operations in program order, which
// the hardware gives (LDS instructions of a wave complete in order); a workgroup barrier here would also wait for
// every outstanding global load and store of the wave.
PT_DEV void wave_sync() { wave_lds_order(); }

PT_DEV int lane_prefix(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// head of the i-th mesh of the (single) TLAS leaf from its LDS copy; root reference and flags made scalar
PT_DEV MeshHead staged_mesh_head(const PairLds &L, int i) {
    const float4 a = L.meshbox[2 * i], b = L.meshbox[2 * i + 1];
    MeshHead h;
    h.bmin = mk3(a.x, a.y, a.z);
    h.bmax = mk3(b.x, b.y, b.z);
    h.root_ref = __builtin_amdgcn_readfirstlane(__float_as_int(a.w));
    const int fm = __builtin_amdgcn_readfirstlane(__float_as_int(b.w)); // (staged as flags | mesh id << 8)
    h.flags = fm & 0xff;
    h.mesh = fm >> 8;
    return h;
}
// the same mesh as a pair sees it, per lane: {root reference, 0, flags, mesh id} (PMODE 2, 4; PMODE 1 has L.meshtab)
PT_DEV int4 staged_mesh_entry(const PairLds &L, int i) {
    const int root = __float_as_int(lds_ld1((const float *)L.meshbox, 8 * i + 3));
    const int fm = __float_as_int(lds_ld1((const float *)L.meshbox, 8 * i + 7));
    return make_int4(root, 0, fm & 0xff, fm >> 8);
}

// Exclusive prefix sum and total of the lanes' leaf sizes (< 32) by bit planes: five ballots and mbcnt pairs, no LDS --
// the shuffle scan it replaces was six dependent ds_bpermute round trips per leaf phase.
PT_DEV void leaf_prefix(int cnt, int &start, int &total) {
    int st = 0, tot = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64((cnt >> b) & 1);
        st += lane_prefix(m) << b;
        tot += __builtin_popcountll(m) << b;
    }
    start = st;
    total = tot;
}

// step 1: root-box tests + ballot/prefix-sum compaction into the LDS pair list
template <bool ANY>
PT_DEV int build_pairs(const KParams &K, const PairLds &L, int lane, bool alive, f3 o, f3 d, float tMax) {
    const RayO w = make_ray(o, d);
    float tE;
    alive = alive && slab(tlas_bmin(K), tlas_bmax(K), w, ANY ? tMax : T_FAR, tE);
    if (ANY)
        L.occ[lane] = 0u;
    else
        L.best[lane] = ~0ull;
    // (the mesh heads come out of LDS, staged at kernel start: read from memory here -- mesh id, then its record, one
    // dependent round trip each -- the loop took ~0.7 us per mesh and ~25 % of a showcase wave's time.  Reading head i + 1
    // while mesh i's box is tested was measured in round 3: Cornell 1.81 -> 1.85 ms, the registers cost more than the round trip)
    int base = 0;
    for (int i = 0; i < K.pair_meshes; ++i) {
        const MeshHead mh = staged_mesh_head(L, i);
        if (ANY && (mh.flags & 2))
            continue;
        bool hb;
        if (mh.flags & 1) {
            float ds;
            const RayO lr = local_ray(K, mh.mesh, w, ds);
            hb = alive && slab(mh.bmin, mh.bmax, lr, ANY ? tMax * ds : T_FAR, tE);
        } else {
            hb = alive && slab(mh.bmin, mh.bmax, w, ANY ? tMax : T_FAR, tE);
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(hb);
        if (hb) {
            ((uint16_t *)L.pairs)[base + lane_prefix(bal)] = (uint16_t)((uint32_t)lane | ((uint32_t)i << 6));
        }
        base += __builtin_popcountll(bal);
    }
    return base;
}

// ray of a pair in the mesh's space (tri_test needs origin and direction only), from the world ray -- fetched from the
// owner lane's registers by ds_bpermute: no ray planes in LDS
// (`GEN`: mt.w is a TLAS index and the rows come from the leaf-order copy; else it is a mesh id)
template <int GEN = 0> PT_DEV void pair_ray_from(const KParams &K, const int4 mt, f3 &o, f3 &d, float &dirScale) {
    dirScale = 1.0f;
    if (mt.z & 1) {
        const float4 *rec = GEN ? (K.tlas_heads + mt.w * TLAS_HEAD_F4) : (K.mesh_recs + mt.w * MESH_REC_F4);
        const f3 lo = xform_point(rec[2], rec[3], rec[4], o);
        const f3 ld = xform_dir(rec[2], rec[3], rec[4], d);
        dirScale = length(ld);
        o = lo;
        d = normalize(ld);
    }
}

PT_DEV Hit closest_hit_pairs(const KParams &K, const PairLds &L, int lane, bool alive, f3 o, f3 d, int &order) {
    const int P = build_pairs<false>(K, L, lane, alive, o, d, T_FAR);
    wave_sync();
    for (int c = 0; c < P; c += 64) {
        // a batch that does not fill the wave (the last one) gives every pair 2^sh lanes, each testing every
        // 2^sh-th triangle: 18 left-over pairs cost 6 iterations instead of 12.  The lanes of a pair merge
        // like the pairs of a ray: 64-bit min on {t, mesh order, triangle} = first minimum in leaf order.
        const int n = (P - c) < 64 ? (P - c) : 64;
        int sh = 0;
        // (not with instanced meshes: their key carries t / dirScale, and two local distances that round to
        // the same world distance must resolve by LOCAL distance first, as one lane's running minimum does)
        while (K.pair_split && (n << (sh + 1)) <= 64 && (2 << sh) <= K.pair_max_leaf)
            ++sh;
        const int p = c + (lane >> sh), sub = lane & ((1 << sh) - 1);
        const bool valid = (lane >> sh) < n;
        const uint32_t e = ((const uint16_t *)L.pairs)[valid ? p : 0];
        const int r = (int)(e & 63u), oi = (int)(e >> 6);
        const int4 mt = L.meshtab[oi];
        // (every lane executes the shuffles: a source lane must be active for ds_bpermute)
        f3 po = mk3(__shfl(o.x, r), __shfl(o.y, r), __shfl(o.z, r));
        f3 pd = mk3(__shfl(d.x, r), __shfl(d.y, r), __shfl(d.z, r));
        float dirScale;
        pair_ray_from<false>(K, mt, po, pd, dirScale);
        RayO pr;
        pr.o = po;
        pr.d = pd;
        float tb = T_FAR;
        int bi = -1;
        // (software-pipelining these LDS reads one packet ahead was measured: 2.82 vs 2.75 ms -- with four
        // waves per SIMD the latency is already covered and the extra live registers cost more)
        const int iters = (K.pair_max_leaf + (1 << sh) - 1) >> sh;
        for (int it = 0; it < iters; ++it) {
            const int i = sub + (it << sh);
            const int slot = mt.x + (i < mt.y ? i : 0);
            const float4 *tp = L.tris + slot * 3 + oi * PAIR_PAD;
            const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
            asm volatile("" ::"v"(p0.w), "v"(p1.w), "v"(p2.w)); // keep the loads ds_read_b128 (b96 is half rate)
            float t, u, v;
            const bool ok = tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), pr, tb, t, u, v);
            if (ok && i < mt.y) {
                tb = t;
                bi = i;
            }
        }
        if (valid && bi >= 0) {
            const float tw = (mt.z & 1) ? tb / dirScale : tb;
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(tw) << 32) | ((unsigned long long)(uint32_t)oi << 16) | (uint32_t)bi;
            __hip_atomic_fetch_min(&L.best[r], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    wave_sync();
    const unsigned long long key = L.best[lane];
    wave_sync(); // the lists are rebuilt by the next trace
    Hit h;
    h.u = h.v = 0.0f;
    order = 0;
    if (!alive || key == ~0ull) {
        h.t = h.t_local = T_FAR;
        h.mesh = -1;
        h.slot = -1;
        return h;
    }
    const int oi = (int)((key >> 16) & 0xffffu), bi = (int)(key & 0xffffu);
    order = oi;
    const int4 mt = L.meshtab[oi];
    h.t = __uint_as_float((uint32_t)(key >> 32));
    h.mesh = mt.w;
    h.slot = mt.x + bi;
    h.t_local = h.t;
    if (mt.z & 1) { // local-space distance of the winner, needed for localPoint (intersection.cuh:382,466)
        const float4 *rec = K.mesh_recs + mt.w * MESH_REC_F4;
        RayO pr;
        pr.o = xform_point(rec[2], rec[3], rec[4], o);
        pr.d = normalize(xform_dir(rec[2], rec[3], rec[4], d));
        const float4 *tp = L.tris + h.slot * 3 + oi * PAIR_PAD;
        const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
        float t, u, v;
        tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), pr, T_FAR, t, u, v);
        h.t_local = t;
    }
    return h;
}

PT_DEV bool any_hit_pairs(const KParams &K, const PairLds &L, int lane, bool alive, f3 o, f3 d, float tMax) {
    const int P = build_pairs<true>(K, L, lane, alive, o, d, tMax);
    // per-lane tMax travels with the ray: reuse the `best` words as a float plane
    float *tmaxv = (float *)L.best;
    tmaxv[lane] = tMax;
    wave_sync();
    for (int c = 0; c < P; c += 64) {
        const int n = (P - c) < 64 ? (P - c) : 64; // 2^sh lanes per pair in a batch that does not fill the wave
        int sh = 0;
        while ((n << (sh + 1)) <= 64 && (2 << sh) <= K.pair_max_leaf)
            ++sh;
        const int p = c + (lane >> sh), sub = lane & ((1 << sh) - 1);
        const bool valid = (lane >> sh) < n;
        const uint32_t e = ((const uint16_t *)L.pairs)[valid ? p : 0];
        const int r = (int)(e & 63u), oi = (int)(e >> 6);
        const int4 mt = L.meshtab[oi];
        // (every lane executes the shuffles: a source lane must be active for ds_bpermute)
        f3 po = mk3(__shfl(o.x, r), __shfl(o.y, r), __shfl(o.z, r));
        f3 pd = mk3(__shfl(d.x, r), __shfl(d.y, r), __shfl(d.z, r));
        float dirScale;
        pair_ray_from<false>(K, mt, po, pd, dirScale);
        RayO pr;
        pr.o = po;
        pr.d = pd;
        float tm = tmaxv[r];
        if (mt.z & 1)
            tm = tm * dirScale;
        bool found = false;
        const int iters = (K.pair_max_leaf + (1 << sh) - 1) >> sh;
        for (int it = 0; it < iters; ++it) {
            const int i = sub + (it << sh);
            const int slot = mt.x + (i < mt.y ? i : 0);
            const float4 *tp = L.tris + slot * 3 + oi * PAIR_PAD;
            const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
            asm volatile("" ::"v"(p0.w), "v"(p1.w), "v"(p2.w)); // keep the loads ds_read_b128 (b96 is half rate)
            float t, u, v;
            const bool ok = tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), pr, tm, t, u, v);
            found |= ok && (i < mt.y);
        }
        if (valid && found)
            L.occ[r] = 1u;
    }
    wave_sync();
    const bool occluded = alive && (L.occ[lane] != 0u);
    wave_sync();
    return occluded;
}

// PMODE 2: the same pair compaction when the BLASes are real trees (single-leaf TLAS): a lane
// takes one (ray, mesh) pair and walks that mesh's BVH with its own LDS stack, so every lane
// traverses SOME mesh instead of idling while the wave walks meshes its ray never touches.
// Within a pair the traversal is the reference's (local best, strict `<`); pairs of one ray
// merge by the same 64-bit min {t bits, mesh order | payload}.  Pair entries are 16 bits here
// ({lane, mesh order < 256}): the list is the second largest LDS user after the stacks.
// PMODE 2 with dynamic refill.  A batch of 64 pairs runs as long as its LONGEST traversal while the
// lanes whose pair missed after three nodes idle (measured on the showcase scene: 17 % of the VALU lanes
// busy).  Here the pair list is a queue: whenever K.fetch_min lanes are idle (or all are), the idle
// lanes take the next pairs -- rank by ballot/mbcnt, `next` is wave-uniform, no atomics -- commit
// their finished pair with the same 64-bit min and start over, while the other lanes keep their
// traversal state.  Every pair is still traversed exactly as before, so the bits cannot change.
// mesh of a pair entry.  GEN = false: `order` indexes the single TLAS leaf (the staged heads).  GEN = true (general TLAS,
// PMODE 3): `order` indexes the TLAS leaf the RAY is currently at (L.leafx[r]), the head comes from the mesh records.
// GEN 0: `order` indexes the single TLAS leaf; 1: leaf slot of this fill | index in that leaf << 2; 2: the TLAS index itself
template <int GEN> PT_DEV int4 pair_mesh(const KParams &K, const PairLds &L, int r, int order) {
    if (!GEN)
        return staged_mesh_entry(L, order);
    // (root reference and flags from the leaf-order copy of the heads; the mesh id only matters for an instance's matrices)
    const int j = GEN == 2 ? order : L.leafx[(order & (TLAS_SLOTS - 1)) * 64 + r] + (order >> 2);
    const int root = __float_as_int(K.tlas_heads[TLAS_HEAD_F4 * j].w), flags = __float_as_int(K.tlas_heads[TLAS_HEAD_F4 * j + 1].w);
    return make_int4(root, 0, flags, j);
}

// one child-pair node: out of the workgroup's LDS copy if it is one of the tree's first TOP_NODES (WG = 4 variant),
// else from memory
PT_DEV void fetch_node(const KParams &K, const PairLds &L, int cur, int local, int order, float4 &n0, float4 &n1, float4 &n2,
                       float4 &n3) {
    if (L.topnodes && (unsigned)local < (unsigned)TOP_NODES) {
        const float4 *q = L.topnodes + (order * TOP_NODES + local) * 4;
        n0 = q[0];
        n1 = q[1];
        n2 = q[2];
        n3 = q[3];
    } else {
        n0 = K.nodes[cur * 4 + 0];
        n1 = K.nodes[cur * 4 + 1];
        n2 = K.nodes[cur * 4 + 2];
        n3 = K.nodes[cur * 4 + 3];
    }
}

// Two steps of the binary near-first walk (bvh_trace_local, intersection.cuh:344-435) from ONE two-level record
// (expand_nodes_kernel): the node's child boxes, then -- if the walk enters an inner child -- that child's child boxes, which
// the reference would test in its next iteration with the same limit `tb` (no leaf is visited in between, so the limit
// cannot have moved).  Pushes are the binary walk's, in its order: the far child with its entry distance (E1), then the far
// grandchild.  Afterwards `cur` is a leaf or a node two levels down; returns true when neither child (or neither
// grandchild) can hold a closer hit: the caller pops.  Same culling, same tie order, half the dependent round trips.
#ifndef PT_TWO_LEVEL
#define PT_TWO_LEVEL 0 // measured (showcase 3.878 vs 3.888 ms, 1 M triangles 1.226 vs 1.200, fluid 0.975 vs 0.983): a build option
#endif
PT_DEV bool descend2(const float4 *__restrict__ nodes2, LdsStack stk, int &sp, const RayO &pr, float tb, int &cur) {
    const float4 *rec = nodes2 + (size_t)cur * NODE2_F4;
    const float4 n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3];
    float tL, tR;
    const bool hL = slab(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), pr, tb, tL);
    const bool hR = slab(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), pr, tb, tR);
    if (!(hL || hR))
        return true;
    const bool nearL = hL && (!hR || tL <= tR);
    const int Lr = __float_as_int(n3.x), Rr = __float_as_int(n3.y);
    if (nearL ? hR : hL) {
        stk.push(sp, nearL ? Rr : Lr, nearL ? tR : tL);
        ++sp;
    }
    cur = nearL ? Lr : Rr;
    if (cur < 0)
        return false;
    const float4 *g = rec + (nearL ? 4 : 8);
    const float4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
    const bool hGL = slab(mk3(g0.x, g0.y, g0.z), mk3(g0.w, g1.x, g1.y), pr, tb, tL);
    const bool hGR = slab(mk3(g1.z, g1.w, g2.x), mk3(g2.y, g2.z, g2.w), pr, tb, tR);
    if (!(hGL || hGR))
        return true;
    const bool nearGL = hGL && (!hGR || tL <= tR);
    const int GLr = __float_as_int(g3.x), GRr = __float_as_int(g3.y);
    if (nearGL ? hGR : hGL) {
        stk.push(sp, nearGL ? GRr : GLr, nearGL ? tR : tL);
        ++sp;
    }
    cur = nearGL ? GLr : GRr;
    return false;
}
// the any-hit walk's two steps (bvh_any_hit_local, intersection.cuh:300-341: no ordering, the right child waits on the stack)
PT_DEV bool descend2_any(const float4 *__restrict__ nodes2, LdsStack stk, int &sp, const RayO &pr, float tm, int &cur) {
    const float4 *rec = nodes2 + (size_t)cur * NODE2_F4;
    const float4 n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3];
    float tL, tR;
    const bool hL = slab(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), pr, tm, tL);
    const bool hR = slab(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), pr, tm, tR);
    if (!(hL || hR))
        return true;
    const int Lr = __float_as_int(n3.x), Rr = __float_as_int(n3.y);
    if (hL && hR) {
        stk.push(sp, Rr, 0.0f);
        ++sp;
    }
    cur = hL ? Lr : Rr;
    if (cur < 0)
        return false;
    const float4 *g = rec + (hL ? 4 : 8);
    const float4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
    const bool hGL = slab(mk3(g0.x, g0.y, g0.z), mk3(g0.w, g1.x, g1.y), pr, tm, tL);
    const bool hGR = slab(mk3(g1.z, g1.w, g2.x), mk3(g2.y, g2.z, g2.w), pr, tm, tR);
    if (!(hGL || hGR))
        return true;
    const int GLr = __float_as_int(g3.x), GRr = __float_as_int(g3.y);
    if (hGL && hGR) {
        stk.push(sp, GRr, 0.0f);
        ++sp;
    }
    cur = hGL ? GLr : GRr;
    return false;
}

// drains the pair queue [0, P): afterwards L.best[r] = min over ray r's pairs of {t bits, order << 24 | slot}
//
// STEAL (option "csteal", PMODE 2): closest-hit subtree stealing WITH VERIFICATION.  A closest-hit phase lasts as long as its
// longest walk -- secondary rays: ~12 node steps per pair, 38-45 wave-iterations per phase at 13-26 % busy lanes
// (profiles/r04_lane_occupancy.txt, by bounce) -- and the long walks are the ones that find nothing to cull with.  Once the
// queue is empty an idle lane takes the BOTTOM entry S of a busy lane's stack with a copy of its ray and of its limit at that
// moment, walks it with its own stack and merges what it finds like a pair of its own.  Why the bits stay:
//  * the bottom entry is what the owner's depth-first walk would visit LAST, so everything the owner still does is the
//    reference's walk with the reference's limits; the reference then enters S with the owner's FINAL limit b (or culls it);
//  * the thief walks S with a limit a >= b (limits only shrink).  As long as every hit a thief accepts lies at or beyond the
//    entry distance of its own leaf box (child boxes lie inside their parents' and the slab arithmetic is monotone, so that
//    is the largest entry distance on its path), the thief's limit stays >= the reference's at every step, it visits a
//    superset of the reference's nodes in the same order, and its closest hit -- if below b -- is the reference's;
//  * the slab test is NOT conservative w.r.t. the triangle test (E3): a thief that accepts a hit closer than its leaf box's
//    entry, and two walks of one (ray, mesh) pair that report the same distance (the reference keeps the one it visits
//    first), mark the ray in L.dirty[0] instead, and the caller traces marked rays again without stealing.
// Victims must have taken K.csteal_min node steps (short walks are not worth a thief: what they leave at the bottom of the
// stack is what their next hit culls).
template <int GEN, bool STEAL = false> PT_DEV void run_closest_queue(const KParams &K, const PairLds &L, int lane, int P, f3 o, f3 d) {
    LdsStack stk{L.stack + lane};
    int next = 0;
    bool busy = false, active = false, xf = false, thief = false;
    int cur = 0, sp = 0, bot = 0, r = 0, oi = 0, sb = -1, rootref = 0, nsteps = 0;
    int gen = 0, vic = 0; // STEAL: walks this lane has begun; a thief's victim: lane | that lane's count at the theft << 8
    float dirScale = 1.0f, tb = T_FAR, tcur = 0.0f; // tcur: entry distance of the box of `cur` (STEAL: the thieves' acceptance test)
    RayO pr = make_ray(mk3(0.0f), mk3(0.0f, 0.0f, 1.0f));
    TravStats ts;
    auto pop = [&]() {
        active = false;
        while (sp > (STEAL ? bot : 0)) {
            --sp;
            int ref;
            float tE;
            stk.pop(sp, ref, tE);
            if (tE < tb) {
                cur = ref;
                tcur = tE;
                active = true;
                break;
            }
        }
    };
    for (;;) {
        TS_WAVE(7);
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!busy);
        const int n_idle = __builtin_popcountll(idle);
        if (next < P && (n_idle >= K.fetch_min || n_idle == 64)) {
            const int p = next + lane_prefix(idle);
            const bool take = !busy && p < P;
            const uint32_t e = ((const uint16_t *)L.pairs)[take ? p : 0];
            // the pair's ray lives in the registers of lane e & 63: every lane executes the shuffles (a source
            // lane must be active for ds_bpermute), the takers keep the result -- no ray planes in LDS
            const int src = (int)(e & 63u);
            f3 po = mk3(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src));
            f3 pd = mk3(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src));
            if (take) {
                r = src;
                oi = (int)(e >> 6);
                const int4 mt = pair_mesh<GEN>(K, L, r, oi);
                pair_ray_from<GEN>(K, mt, po, pd, dirScale);
                pr = make_ray(po, pd);
                xf = (mt.z & 1) != 0;
                cur = rootref = mt.x;
                sp = bot = 0;
                nsteps = 0;
                thief = false;
                ++gen;
                tb = T_FAR;
                sb = -1;
                busy = active = true;
            }
            next += n_idle;
        }
        if (!__builtin_amdgcn_ballot_w64(busy))
            break;
        bool can_steal = false;
        if (STEAL && next >= P && K.csteal_follow) {
            // A thief follows its victim's limit: the victim's walk comes BEFORE the stolen subtree in the reference's order, so
            // whatever it has found by now the reference has found by the time it enters the subtree -- as long as the victim
            // is still on the walk it was robbed on (its count of walks begun).
            const int vl = vic & 63;
            const float vt = __shfl(tb, vl);
            const int vg = __shfl(gen, vl);
            if (busy && thief && vg == (vic >> 8) && vt < tb) {
                tb = vt;
                sb = -1; // (what the thief had found lies behind it)
            }
        }
        if (STEAL && next >= P) {
            const unsigned long long thieves = __builtin_amdgcn_ballot_w64(!busy);
            const bool is_victim0 = busy && active && sp > bot && nsteps >= K.csteal_min;
            const unsigned long long victims = __builtin_amdgcn_ballot_w64(is_victim0);
            if (thieves && victims) {
                const int nt = __builtin_popcountll(thieves), nv = __builtin_popcountll(victims);
                const int k = nt < nv ? nt : nv;
                const int vrank = lane_prefix(victims), trank = lane_prefix(thieves);
                const bool is_victim = is_victim0 && vrank < k;
                if (is_victim)
                    L.owner[vrank] = (unsigned char)lane;
                wave_lds_order();
                const bool steal = !busy && trank < k;
                const int v = steal ? (int)L.owner[trank] : lane;
                const int vb = __shfl(bot, v);
                RayO npr;
                npr.o = mk3(__shfl(pr.o.x, v), __shfl(pr.o.y, v), __shfl(pr.o.z, v));
                npr.d = mk3(__shfl(pr.d.x, v), __shfl(pr.d.y, v), __shfl(pr.d.z, v));
                npr.inv = mk3(__shfl(pr.inv.x, v), __shfl(pr.inv.y, v), __shfl(pr.inv.z, v));
                const float ntb = __shfl(tb, v), nds = __shfl(dirScale, v);
                const int nr = __shfl(r, v), noi = __shfl(oi | (xf ? 1 << 30 : 0), v), ng = __shfl(gen, v);
                if (steal) {
                    const uint2 e = L.stack[vb * 64 + v];
                    const float tE = __uint_as_float(e.y);
                    if (tE < ntb) { // (else the reference culls it as well: its limit is at most the victim's current one)
                        cur = (int)e.x;
                        tcur = tE;
                        npr.sx = npr.inv.x < 0;
                        npr.sy = npr.inv.y < 0;
                        npr.sz = npr.inv.z < 0;
                        pr = npr;
                        tb = ntb;
                        dirScale = nds;
                        r = nr;
                        oi = noi & ~(1 << 30);
                        xf = (noi >> 30) & 1;
                        rootref = -(1 << 30); // (a stolen subtree lies below the staged levels: its nodes come from memory)
                        sp = bot = 0;
                        nsteps = 0;
                        sb = -1;
                        ++gen;
                        vic = v | (ng << 8);
                        thief = true;
                        busy = active = true;
                        TS_EVENT(0);
                    }
                }
                if (is_victim)
                    ++bot;
                wave_lds_order();
            }
            // (the node loop below yields every K.csteal steps only while a lane is idle AND a walk is, or soon will be, worth stealing from)
            can_steal = thieves != 0ull && __builtin_amdgcn_ballot_w64(busy && active && nsteps + K.csteal >= K.csteal_min) != 0ull;
        }
        // Inner nodes.  A lane needs ~2 node steps to its next leaf, the slowest of 64 needs ~12: the
        // descent stops as soon as K.leaf_min lanes wait at a leaf (wave-uniform loop, predicated step, so
        // the waiting lanes take part in the ballots); they are served below, pop, and rejoin it.
        const unsigned long long t_nd = TS_NOW();
        int steps = 0;
        for (;;) {
            const bool innode = active && cur >= 0;
            if (!__builtin_amdgcn_ballot_w64(innode))
                break;
            // (with thieves at work few lanes idle through a descent, so it pays to gather more leaves per leaf phase)
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(active && cur < 0)) >= (STEAL ? K.csteal_leaf_min : K.leaf_min))
                break;
            if (STEAL && can_steal && ++steps > K.csteal) // idle lanes are waiting for stack entries to take
                break;
#ifdef PT_TRAV_STATS
            {   // what the lanes that take no node step in this wave-iteration are doing: waiting at a leaf / without a walk
                const unsigned long long wl = __builtin_amdgcn_ballot_w64(active && cur < 0), il = __builtin_amdgcn_ballot_w64(!active);
                if (lane == 0) {
                    atomicAdd(&g_trav_dbg[1025 + 4], (unsigned long long)__builtin_popcountll(wl));
                    atomicAdd(&g_trav_dbg[1025 + 5], (unsigned long long)__builtin_popcountll(il));
                }
            }
#endif
            if (innode) {
                TS_WAVE(2);
                TS_LANE(3);
                if (STEAL)
                    ++nsteps;
                if (PT_TWO_LEVEL && !L.topnodes && !STEAL) {
                    if (descend2(K.nodes2, stk, sp, pr, tb, cur))
                        pop();
                    continue;
                }
                float4 n0, n1, n2, n3;
                fetch_node(K, L, cur, cur - rootref, oi, n0, n1, n2, n3);
                float tL, tR;
                const bool hL = slab(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), pr, tb, tL);
                const bool hR = slab(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), pr, tb, tR);
                const int Lr = __float_as_int(n3.x), Rr = __float_as_int(n3.y);
                if (hL || hR) {
                    const bool nearL = hL && (!hR || tL <= tR);
                    if (nearL ? hR : hL) {
                        stk.push(sp, nearL ? Rr : Lr, nearL ? tR : tL);
                        ++sp;
                    }
                    cur = nearL ? Lr : Rr;
                    if (STEAL)
                        tcur = nearL ? tL : tR;
                } else {
                    pop();
                }
            }
        }
        TS_ADDLQ(9, t_nd);
        const unsigned long long t_lf = TS_NOW();
        const bool atleaf = active && cur < 0;
        if (STEAL && !__builtin_amdgcn_ballot_w64(atleaf)) {
            // (a yield with no lane at a leaf: nothing to test)
        } else
        if (K.leaf_pairs) {
            // Leaf phase as (lane, triangle) pairs.  Lane by lane it runs as long as the largest leaf (<= 17
            // tests) with the lanes that are not at a leaf idle: 21 % of the lanes busy on the showcase scene.
            // Instead the tests of all waiting lanes form one list (prefix sum of the leaf sizes), 64 tests
            // per iteration; a test takes its ray from LDS and merges into its lane with a 64-bit LDS min on
            // {t bits, index in leaf} = smallest t, then first index: the sequential loop's strict `<`.
            int cnt = 0, first = 0;
            if (atleaf) {
                const int2 lf = K.leaves[~cur];
                first = lf.x;
                cnt = lf.y;
            }
            int start, T;
            leaf_prefix(cnt, start, T);
            if (atleaf) {
                L.lkey[lane] = ~0ull;
                for (int i = 0; i < cnt; ++i)
                    L.owner[start + i] = (unsigned char)lane;
            }
            wave_lds_order();
            for (int j0 = 0; j0 < T; j0 += 64) {
                // everything a test needs of its lane comes out of that lane's registers (ds_bpermute, all
                // lanes executing), so the list costs LDS only for `owner` and the merge keys
                const int j = j0 + lane;
                const int o = L.owner[j < T ? j : 0];
                const int i = j - __shfl(start, o);
                const int slot = __shfl(first, o) + i;
                const float tl = __shfl(tb, o);
                RayO tr;
                tr.o = mk3(__shfl(pr.o.x, o), __shfl(pr.o.y, o), __shfl(pr.o.z, o));
                tr.d = mk3(__shfl(pr.d.x, o), __shfl(pr.d.y, o), __shfl(pr.d.z, o));
                if (j < T) {
                    TS_WAVE(5);
                    TS_LANE(6);
                    const float4 *tp = K.tris + (size_t)slot * 3;
                    const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
                    float t, u, v;
                    if (tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), tr, tl, t, u, v))
                        __hip_atomic_fetch_min(&L.lkey[o], ((unsigned long long)__float_as_uint(t) << 32) | (uint32_t)i,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            wave_lds_order();
            if (atleaf) {
                TS_WAVE(4);
                const unsigned long long key = L.lkey[lane];
                if (key != ~0ull) {
                    tb = __uint_as_float((uint32_t)(key >> 32));
                    sb = first + (int)(uint32_t)key;
                    if (STEAL && thief && !(tcur < tb)) { // a hit in front of its own leaf box: the reference may never have come here
                        __hip_atomic_fetch_or(L.dirty, 1ull << r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        TS_EVENT(1);
                    }
                }
                pop();
            }
        } else
        if (atleaf) { // cur is a leaf
            TS_WAVE(4);
            const int2 lf = K.leaves[~cur];
            const float4 *tp = K.tris + (size_t)lf.x * 3;
            float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
            if (lf.y > 0) {
                p0 = tp[0];
                p1 = tp[1];
                p2 = tp[2];
            }
            bool hit_here = false;
            for (int i = 0; i < lf.y; ++i) {
                TS_WAVE(5);
                TS_LANE(6);
                const int nx = (i + 1 < lf.y) ? (i + 1) : i;
                const float4 q0 = tp[nx * 3 + 0], q1 = tp[nx * 3 + 1], q2 = tp[nx * 3 + 2];
                float t, u, v;
                if (tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), pr, tb, t, u, v)) {
                    tb = t;
                    sb = lf.x + i;
                    hit_here = true;
                }
                p0 = q0;
                p1 = q1;
                p2 = q2;
            }
            if (STEAL && thief && hit_here && !(tcur < tb))
                __hip_atomic_fetch_or(L.dirty, 1ull << r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pop();
        }
        TS_ADDLQ(10, t_lf);
        if (busy && !active) { // this pair is finished: merge it into its ray
            if (sb >= 0) {
                const float tw = xf ? tb / dirScale : tb;
                // (GEN: one minimum per leaf slot of the fill; the tie order is the index within that leaf)
                const int ko = GEN ? (oi >> 2) : oi, kb = GEN ? (oi & (TLAS_SLOTS - 1)) * 64 + r : r;
                const unsigned long long key =
                    ((unsigned long long)__float_as_uint(tw) << 32) | ((unsigned long long)(uint32_t)ko << 24) | (uint32_t)sb;
                const unsigned long long old = __hip_atomic_fetch_min(&L.best[kb], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // (two walks of ONE pair at the same distance: the reference keeps the one it visits first, the minimum the
                // lower slot)
                if (STEAL && (old >> 24) == (key >> 24) && old != key) {
                    __hip_atomic_fetch_or(L.dirty, 1ull << r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    TS_EVENT(2);
                    TS_DBG(old, key, (unsigned long long)r | ((unsigned long long)lane << 8) | ((unsigned long long)thief << 16) | ((unsigned long long)(unsigned)nsteps << 32),
                           (unsigned long long)__float_as_uint(tcur) | ((unsigned long long)(unsigned)L.stat_bounce << 32));
                }
            }
            busy = false;
        }
    }
#ifdef PT_TRAV_STATS
    ts.v[0] = lane == 0 ? 1u : 0u;
    ts.v[1] = lane == 0 ? (unsigned)P : 0u;
#endif
    ts.flush(0, lane, L.stat_bounce);
}

// local-space distance of the winner in an instanced mesh, needed for localPoint (intersection.cuh:382,466)
PT_DEV float winner_t_local(const KParams &K, int mesh, int slot, f3 o, f3 d) {
    const float4 *rec = K.mesh_recs + mesh * MESH_REC_F4;
    RayO lr;
    lr.o = xform_point(rec[2], rec[3], rec[4], o);
    lr.d = normalize(xform_dir(rec[2], rec[3], rec[4], d));
    const float4 p0 = K.tris[slot * 3 + 0], p1 = K.tris[slot * 3 + 1], p2 = K.tris[slot * 3 + 2];
    float t, u, v;
    tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), lr, T_FAR, t, u, v);
    return t;
}

PT_DEV Hit closest_hit_pairs_dyn(const KParams &K, const PairLds &L, int lane, bool alive, f3 o, f3 d) {
    const int P = build_pairs<false>(K, L, lane, alive, o, d, T_FAR);
    const bool stealing = K.csteal > 0 && L.dirty; // (wave-uniform)
    if (stealing && lane == 0)
        L.dirty[0] = 0ull;
    wave_sync();
    const unsigned long long t_q = TS_NOW();
    if (stealing)
        run_closest_queue<false, true>(K, L, lane, P, o, d);
    else
        run_closest_queue<false>(K, L, lane, P, o, d);
    TS_ADDLQ(8, t_q);
    wave_sync();
    if (stealing) {
        // rays a thief could not vouch for (run_closest_queue): traced again, every pair walked by one lane
        const unsigned long long dm = L.dirty[0];
        wave_sync();
        if (dm) {
#ifdef PT_TRAV_STATS
            if (lane == 0)
                atomicAdd(&g_trav_dbg[1025 + 3], (unsigned long long)__builtin_popcountll(dm));
#endif
            const unsigned long long keep = L.best[lane];
            wave_sync();
            const bool redo = alive && ((dm >> lane) & 1ull);
            const int P2 = build_pairs<false>(K, L, lane, redo, o, d, T_FAR);
            if (!redo) // (build_pairs resets every lane's minimum: the other rays' go back)
                L.best[lane] = keep;
            wave_sync();
            run_closest_queue<false>(K, L, lane, P2, o, d);
            wave_sync();
        }
    }
    const unsigned long long key = L.best[lane];
    wave_sync();
    Hit h;
    h.u = h.v = 0.0f;
    if (!alive || key == ~0ull) {
        h.t = h.t_local = T_FAR;
        h.mesh = -1;
        h.slot = -1;
        return h;
    }
    const int4 mt = staged_mesh_entry(L, (int)((key >> 24) & 0xffu));
    h.t = __uint_as_float((uint32_t)(key >> 32));
    h.mesh = mt.w;
    h.slot = (int)(key & 0xffffffu);
    h.t_local = h.t;
    if (mt.z & 1)
        h.t_local = winner_t_local(K, h.mesh, h.slot, o, d);
    return h;
}

// Any hit, same queue.  A pair whose ray is already known to be occluded is dropped at refill (the
// answer is an OR over the ray's pairs).
// drains the pair queue [0, P): afterwards L.occ[r] != 0 for every ray r one of whose pairs found a hit
template <int GEN> PT_DEV void run_any_queue(const KParams &K, const PairLds &L, int lane, int P, f3 o, f3 d) {
    LdsStack stk{L.stack + lane};
    const float *tmaxv = (const float *)L.best;
    int next = 0;
    bool busy = false;
    int cur = 0, sp = 0, bot = 0, r = 0, oi = 0, rootref = 0; // the lane's stack is entries [bot, sp): thieves take from the bottom
    float tm = 0.0f;
    RayO pr = make_ray(mk3(0.0f), mk3(0.0f, 0.0f, 1.0f));
    TravStats ts;
    auto pop = [&]() {
        busy = false;
        if (sp > bot && L.occ[r] == 0u) { // (nothing left to find for a ray some lane has already seen blocked)
            --sp;
            float tE;
            stk.pop(sp, cur, tE);
            busy = true;
        }
    };
    for (;;) {
        TS_WAVE(7);
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!busy);
        const int n_idle = __builtin_popcountll(idle);
        if (next < P && (n_idle >= K.fetch_min || n_idle == 64)) {
            const int p = next + lane_prefix(idle);
            const bool take = !busy && p < P;
            const uint32_t e = ((const uint16_t *)L.pairs)[take ? p : 0];
            const int src = (int)(e & 63u); // the ray comes out of its owner lane's registers, see run_closest_queue
            f3 po = mk3(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src));
            f3 pd = mk3(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src));
            if (take) {
                r = src;
                if (L.occ[r] == 0u) {
                    oi = (int)(e >> 6);
                    const int4 mt = pair_mesh<GEN>(K, L, r, oi);
                    float dirScale;
                    pair_ray_from<GEN>(K, mt, po, pd, dirScale);
                    pr = make_ray(po, pd);
                    tm = tmaxv[r];
                    if (mt.z & 1)
                        tm = tm * dirScale;
                    cur = rootref = mt.x;
                    sp = bot = 0;
                    busy = true;
                }
            }
            next += n_idle;
        }
        if (!__builtin_amdgcn_ballot_w64(busy))
            break;
        // ---- work stealing.  A wave has ~10 shadow rays (showcase scene) and each walks ~18 nodes one after
        // the other: 6 % of the lanes busy.  Any-hit is an OR over the subtrees, in any order (E4), so once the
        // pair queue is empty an idle lane takes the BOTTOM entry of a busy lane's stack -- the largest
        // subtree that lane still owes -- together with a copy of its ray (ds_bpermute), and walks it with
        // its own stack.  Stacks only ever lose entries this way, so their depth bound is untouched.
        bool can_steal = false;
        if (K.steal && next >= P) {
            const unsigned long long thieves = __builtin_amdgcn_ballot_w64(!busy);
            const unsigned long long victims = __builtin_amdgcn_ballot_w64(busy && sp > bot);
            if (thieves && victims) {
                const int nt = __builtin_popcountll(thieves), nv = __builtin_popcountll(victims);
                const int k = nt < nv ? nt : nv;
                const bool is_victim = busy && sp > bot;
                const int vrank = lane_prefix(victims), trank = lane_prefix(thieves);
                if (is_victim && vrank < k)
                    L.owner[vrank] = (unsigned char)lane;
                wave_lds_order();
                const bool steal = !busy && trank < k;
                const int v = steal ? (int)L.owner[trank] : lane;
                const int vb = __shfl(bot, v);
                RayO npr;
                npr.o = mk3(__shfl(pr.o.x, v), __shfl(pr.o.y, v), __shfl(pr.o.z, v));
                npr.d = mk3(__shfl(pr.d.x, v), __shfl(pr.d.y, v), __shfl(pr.d.z, v));
                npr.inv = mk3(__shfl(pr.inv.x, v), __shfl(pr.inv.y, v), __shfl(pr.inv.z, v));
                const float ntm = __shfl(tm, v);
                const int nr = __shfl(r, v);
                if (steal) {
                    const uint2 e = L.stack[vb * 64 + v];
                    cur = (int)e.x;
                    npr.sx = npr.inv.x < 0;
                    npr.sy = npr.inv.y < 0;
                    npr.sz = npr.inv.z < 0;
                    pr = npr;
                    tm = ntm;
                    r = nr;
                    rootref = -(1 << 30); // (a stolen subtree lies below the staged levels: its nodes come from memory)
                    sp = bot = 0;
                    busy = true;
                }
                if (is_victim && vrank < k)
                    ++bot;
                wave_lds_order();
            }
            can_steal = (thieves != 0ull);
        }
        int steps = 0;
        for (;;) { // inner nodes, see closest_hit_pairs_dyn
            const bool innode = busy && cur >= 0;
            if (!__builtin_amdgcn_ballot_w64(innode))
                break;
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(busy && cur < 0)) >= K.leaf_min)
                break;
            if (can_steal && ++steps > K.steal) // idle lanes are waiting for new stack entries to take
                break;
            if (innode) {
                TS_WAVE(2);
                TS_LANE(3);
                if (PT_TWO_LEVEL && !L.topnodes) {
                    if (descend2_any(K.nodes2, stk, sp, pr, tm, cur))
                        pop();
                    continue;
                }
                float4 n0, n1, n2, n3;
                fetch_node(K, L, cur, cur - rootref, oi, n0, n1, n2, n3);
                float tL, tR;
                const bool hL = slab(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), pr, tm, tL);
                const bool hR = slab(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), pr, tm, tR);
                const int Lr = __float_as_int(n3.x), Rr = __float_as_int(n3.y);
                if (hL && hR) {
                    stk.push(sp, Rr, 0.0f);
                    ++sp;
                    cur = Lr;
                } else if (hL || hR) {
                    cur = hL ? Lr : Rr;
                } else {
                    pop();
                }
            }
        }
        const bool atleaf = busy && cur < 0;
        if (K.leaf_pairs) { // leaf phase as (lane, triangle) pairs, see closest_hit_pairs_dyn; any hit: a flag per lane
            int cnt = 0, first = 0;
            if (atleaf) {
                const int2 lf = K.leaves[~cur];
                first = lf.x;
                cnt = lf.y;
            }
            int start, T;
            leaf_prefix(cnt, start, T);
            if (atleaf) {
                L.lkey[lane] = 0ull;
                for (int i = 0; i < cnt; ++i)
                    L.owner[start + i] = (unsigned char)lane;
            }
            wave_lds_order();
            for (int j0 = 0; j0 < T; j0 += 64) {
                const int j = j0 + lane;
                const int o = L.owner[j < T ? j : 0];
                const int i = j - __shfl(start, o);
                const int slot = __shfl(first, o) + i;
                const float tl = __shfl(tm, o);
                RayO tr;
                tr.o = mk3(__shfl(pr.o.x, o), __shfl(pr.o.y, o), __shfl(pr.o.z, o));
                tr.d = mk3(__shfl(pr.d.x, o), __shfl(pr.d.y, o), __shfl(pr.d.z, o));
                if (j < T) {
                    TS_WAVE(5);
                    TS_LANE(6);
                    const float4 *tp = K.tris + (size_t)slot * 3;
                    const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
                    float t, u, v;
                    if (tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), tr, tl, t, u, v))
                        L.lkey[o] = 1ull;
                }
            }
            wave_lds_order();
            if (atleaf) {
                TS_WAVE(4);
                if (L.lkey[lane] != 0ull) {
                    L.occ[r] = 1u;
                    busy = false;
                } else {
                    pop();
                }
            }
        } else
        if (atleaf) { // cur is a leaf
            TS_WAVE(4);
            const int2 lf = K.leaves[~cur];
            bool found = false;
            for (int i = 0; i < lf.y; ++i) {
                TS_WAVE(5);
                TS_LANE(6);
                const int slot = lf.x + i;
                const float4 p0 = K.tris[slot * 3 + 0], p1 = K.tris[slot * 3 + 1], p2 = K.tris[slot * 3 + 2];
                float t, u, v;
                found |= tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), pr, tm, t, u, v);
            }
            if (found) {
                L.occ[r] = 1u;
                busy = false;
            } else {
                pop();
            }
        }
    }
#ifdef PT_TRAV_STATS
    ts.v[0] = lane == 0 ? 1u : 0u;
    ts.v[1] = lane == 0 ? (unsigned)P : 0u;
#endif
    ts.flush(8, lane, L.stat_bounce);
}

PT_DEV bool any_hit_pairs_dyn(const KParams &K, const PairLds &L, int lane, bool alive, f3 o, f3 d, float tMax) {
    const int P = build_pairs<true>(K, L, lane, alive, o, d, tMax);
    ((float *)L.best)[lane] = tMax;
    wave_sync();
    const unsigned long long t_q = TS_NOW();
    run_any_queue<false>(K, L, lane, P, o, d);
    TS_ADDLQ(11, t_q);
    wave_sync();
    const bool occluded = alive && (L.occ[lane] != 0u);
    wave_sync();
    return occluded;
}

// ---------------------------------------------------------------------------------
// PMODE 3: a real TLAS (more meshes than one TLAS leaf holds) through the same pair machinery, in ROUNDS.
// traceRay (intersection.cuh:526-605) walks the TLAS near child first and culls nodes with the closest hit so
// far; at a leaf it traces every mesh from scratch (bvh_trace restarts at 1e30) and keeps strict `<` minima in
// leaf order.  Here every lane walks the TLAS itself (a handful of nodes, stack in LDS) up to its next leaf; then
// the wave turns the lanes' leaves into (ray, mesh) pairs -- index within the ray's own leaf as the tie order --
// and drains them with run_closest_queue; every lane merges its leaf's minimum with strict `<` and walks on with
// the updated limit.  So a mesh is traced iff the reference traces it.  Shadow rays do the same with the any-hit
// walk (bvh_any_hit_tlas, intersection.cuh:481-524) and stop at the first blocked round.
#ifndef PT_ROOT_CHUNK
#define PT_ROOT_CHUNK 2
#endif
// Root-box tests of every lane's NEXT leaf (`slot`-th leaf of this fill) -> pair entries lane | slot << 6 | index << 8
// appended at `base`; returns the new length of the list.
template <bool ANY> PT_DEV uint32_t root_masks(const KParams &K, int lane, bool has, int2 lf, const RayO &w, float tMax) {
    // Pass 1, lock-step over the leaf's entries: ONE world-space slab test per entry.  For an untransformed mesh it is the
    // reference's root-box test.  For an instance it is a conservative pre-test: its first-pass box (gather_tlas_heads_
    // kernel), grown by inst_c2 * |o|_1 for this ray, contains every ray the reference's local-space test can accept, so a
    // miss here is a miss there; a hit only makes the entry a candidate.  Pass 2: every lane runs the reference's test
    // (ray into the instance's space: two transforms, a normalisation, three divisions, ~160 VALU) on ITS candidates,
    // one per iteration whatever their positions in the leaf.  (One lock-step loop that transformed the ray for every
    // instance entry of every lane's leaf was 60 % of a PMODE 3 frame.)
    uint32_t mask = 0u, inst = 0u;
    const float grow = K.inst_c2 * (__builtin_fabsf(w.o.x) + __builtin_fabsf(w.o.y) + __builtin_fabsf(w.o.z));
    for (int i0 = 0; i0 < K.tlas_max_leaf; i0 += PT_ROOT_CHUNK) {
        float4 ha[PT_ROOT_CHUNK], hb4[PT_ROOT_CHUNK];
#pragma unroll
        for (int k = 0; k < PT_ROOT_CHUNK; ++k) {
            const bool in = has && (i0 + k) < lf.y;
            const int j = in ? lf.x + i0 + k : 0;
            ha[k] = K.tlas_heads[TLAS_HEAD_F4 * j];
            hb4[k] = K.tlas_heads[TLAS_HEAD_F4 * j + 1];
        }
#pragma unroll
        for (int k = 0; k < PT_ROOT_CHUNK; ++k) {
            const int flags = __float_as_int(hb4[k].w);
            const bool in = has && (i0 + k) < lf.y && !(ANY && (flags & 2));
            const bool is_inst = (flags & 1) != 0;
            const float g = is_inst ? grow : 0.0f; // (x - 0 and x + 0 are x: an untransformed mesh's box is tested as it is)
            float tE;
            const bool hb = slab(mk3(ha[k].x - g, ha[k].y - g, ha[k].z - g), mk3(hb4[k].x + g, hb4[k].y + g, hb4[k].z + g), w,
                                 ANY ? (is_inst ? tMax * 1.0001f + g : tMax) : T_FAR, tE);
            mask |= (in && !is_inst && hb) ? (1u << (i0 + k)) : 0u;
            inst |= (in && is_inst && hb) ? (1u << (i0 + k)) : 0u;
        }
        if (!__builtin_amdgcn_ballot_w64(has && i0 + PT_ROOT_CHUNK < lf.y))
            break;
    }
    while (__builtin_amdgcn_ballot_w64(inst != 0u)) {
        if (inst != 0u) {
            const int i = __builtin_ctz(inst);
            inst &= inst - 1u;
            const float4 *rec = K.tlas_heads + TLAS_HEAD_F4 * (lf.x + i);
            const float4 a = rec[5], b = rec[6];
            float ds, tE;
            const RayO lr = local_ray_rows(rec[2], rec[3], rec[4], w, ds);
            if (slab(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), lr, ANY ? tMax * ds : T_FAR, tE))
                mask |= 1u << i;
        }
    }
    return mask;
}
// the k-th hit of every lane, k = 0, 1, ...: ballot / prefix-sum compaction into the pair list (a leaf holds at most 17
// meshes of which a ray's root tests pass one or two); entry = lane | code(i) << 6 for hit i of the lane's leaf
template <class Code> PT_DEV int append_pairs(const PairLds &L, int lane, uint32_t mask, int base, Code code) {
    for (;;) {
        const bool more = mask != 0u;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(more);
        if (!bal)
            break;
        if (more) {
            const int i = __builtin_ctz(mask);
            mask &= mask - 1u;
            ((uint16_t *)L.pairs)[base + lane_prefix(bal)] = (uint16_t)((uint32_t)lane | ((uint32_t)code(i) << 6));
        }
        base += __builtin_popcountll(bal);
    }
    return base;
}
template <bool ANY>
PT_DEV int build_pairs_general(const KParams &K, const PairLds &L, int lane, bool has, int2 lf, const RayO &w, float tMax, int slot,
                               int base) {
    const uint32_t mask = root_masks<ANY>(K, lane, has, lf, w, tMax);
    return append_pairs(L, lane, mask, base, [slot](int i) { return slot | (i << 2); });
}

// One fill: up to TLAS_SLOTS times every lane walks the TLAS to its next leaf (near child first, culling with the closest
// hit it knows) and the leaf's root-box tests append pairs -- until the list holds TLAS_FILL_TARGET pairs or no lane
// can advance.  The second and later leaves of a fill are walked with a limit that the first one may still lower:
// a SUPERSET of what the reference visits, in the same order (the near/far order of two children that are both hit
// does not depend on the limit).  Child boxes lie inside their parent's (exact min/max unions), and the slab
// arithmetic is monotone in the box, so a node's entry distance is never below its ancestors': the reference visits a
// leaf of that superset iff the leaf's own entry distance is below the limit at that moment.  After the queue run the
// fill's leaves are therefore replayed in order: entry < best ? merge its minimum with strict `<` : skip it.  Stack
// entries the reference would not have pushed fail the same test when they are popped.
PT_DEV Hit closest_hit_pairs_tlas(const KParams &K, const PairLds &L, int lane, bool alive, f3 o, f3 d, CycleAcc &cyc) {
    const RayO w = make_ray(o, d);
    float tE;
    bool t_active = alive && slab(tlas_bmin(K), tlas_bmax(K), w, T_FAR, tE);
    Hit best;
    best.t = best.t_local = T_FAR;
    best.u = best.v = 0.0f;
    best.mesh = best.slot = -1;
    int tcur = K.tlas_root_ref, tsp = 0;
    float tcur_e = 0.0f; // entry distance of the node `tcur`
    bool need_pop = false;
    auto pop_t = [&]() { // the next TLAS subtree that can still hold a closer hit (E1)
        t_active = false;
        while (tsp > 0) {
            --tsp;
            const uint2 e = L.tstack[tsp * 64 + lane];
            if (__uint_as_float(e.y) < best.t) {
                tcur = (int)e.x;
                tcur_e = __uint_as_float(e.y);
                t_active = true;
                break;
            }
        }
    };
    for (;;) {
        int base = 0, ns = 0;
        float se[TLAS_SLOTS];
        int sx[TLAS_SLOTS];
#pragma unroll
        for (int k = 0; k < TLAS_SLOTS; ++k) {
            se[k] = 0.0f;
            sx[k] = 0;
        }
#pragma unroll
        for (int step = 0; step < TLAS_SLOTS; ++step) {
            int2 lf = make_int2(0, 0);
            bool has = false;
            const unsigned long long t_walk = TS_NOW();
            if (need_pop)
                pop_t();
            while (t_active && !has) {
                if (tcur >= 0) {
                    const float4 n0 = K.tlas_nodes[tcur * 4 + 0], n1 = K.tlas_nodes[tcur * 4 + 1], n2 = K.tlas_nodes[tcur * 4 + 2],
                                 n3 = K.tlas_nodes[tcur * 4 + 3];
                    float tL, tR;
                    const bool hL = slab(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), w, best.t, tL);
                    const bool hR = slab(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), w, best.t, tR);
                    const int Lr = __float_as_int(n3.x), Rr = __float_as_int(n3.y);
                    if (hL || hR) {
                        const bool nearL = hL && (!hR || tL <= tR);
                        if (nearL ? hR : hL) {
                            L.tstack[tsp * 64 + lane] = make_uint2((uint32_t)(nearL ? Rr : Lr), __float_as_uint(nearL ? tR : tL));
                            ++tsp;
                        }
                        tcur = nearL ? Lr : Rr;
                        tcur_e = nearL ? tL : tR;
                    } else {
                        pop_t();
                    }
                } else {
                    lf = K.tlas_leaves[~tcur];
                    has = true;
                }
            }
            need_pop = has;
            TS_ADDQ(9, t_walk);
            // a lane's k-th leaf of this fill sits in slot k (lanes that found none keep their count)
            const int slot = ns;
            if (has) {
                L.leafx[slot * 64 + lane] = lf.x;
                L.best[slot * 64 + lane] = ~0ull;
#pragma unroll
                for (int k = 0; k < TLAS_SLOTS; ++k)
                    if (k == slot) {
                        se[k] = tcur_e;
                        sx[k] = lf.x;
                    }
                ++ns;
            }
            if (!__builtin_amdgcn_ballot_w64(has))
                break;
            const unsigned long long t_b = TS_NOW();
            base = build_pairs_general<false>(K, L, lane, has, lf, w, T_FAR, slot, base);
            TS_ADDQ(10, t_b);
            if (base >= TLAS_FILL_TARGET)
                break;
        }
        if (!__builtin_amdgcn_ballot_w64(ns > 0))
            break;
        wave_sync();
        const unsigned long long t_q = TS_NOW();
        run_closest_queue<true>(K, L, lane, base, o, d);
        wave_sync();
        TS_ADDQ(8, t_q);
#ifdef PT_TRAV_STATS
        cyc.c[3] += 1; // fills (closest)
#endif
#pragma unroll
        for (int k = 0; k < TLAS_SLOTS; ++k) {
            if (k < ns && (k == 0 || se[k] < best.t)) { // (the first leaf of a fill was reached with the exact limit)
                const unsigned long long key = L.best[k * 64 + lane];
                if (key != ~0ull) {
                    const float t = __uint_as_float((uint32_t)(key >> 32));
                    if (t < best.t) { // strict <: an earlier leaf keeps a tie (intersection.cuh:561)
                        best.t = t;
                        best.mesh = K.tlas_mesh_ids[sx[k] + (int)((key >> 24) & 0xffu)];
                        best.slot = (int)(key & 0xffffffu);
                    }
                }
            }
        }
        wave_sync();
    }
    best.t_local = best.t;
    if (best.mesh >= 0 && (__float_as_int(K.mesh_recs[best.mesh * MESH_REC_F4 + 1].w) & 1))
        best.t_local = winner_t_local(K, best.mesh, best.slot, o, d);
    return best;
}

// One leaf per ray and fill (the pair entry names the leaf slot and the index in it): what scenes with more than 1024 meshes
// use, whose TLAS indices do not fit the 16-bit pair entries of the multi-leaf fills below.
PT_DEV bool any_hit_pairs_tlas_rounds(const KParams &K, const PairLds &L, int lane, bool alive, f3 o, f3 d, float tMax, CycleAcc &cyc) {
    const RayO w = make_ray(o, d);
    float tE;
    bool t_active = alive && slab(tlas_bmin(K), tlas_bmax(K), w, tMax, tE);
    ((float *)L.best)[lane] = tMax;
    L.occ[lane] = 0u;
    int tcur = K.tlas_root_ref, tsp = 0;
    bool need_pop = false;
    auto pop_t = [&]() {
        t_active = false;
        if (tsp > 0) {
            --tsp;
            tcur = (int)L.tstack[tsp * 64 + lane].x;
            t_active = true;
        }
    };
    wave_sync();
    for (;;) {
        int base = 0;
        bool any_leaf = false;
#pragma unroll
        for (int step = 0; step < TLAS_SLOTS; ++step) {
            int2 lf = make_int2(0, 0);
            bool has = false;
            const unsigned long long t_walk = TS_NOW();
            if (need_pop)
                pop_t();
            while (t_active && !has) {
                if (tcur >= 0) {
                    const float4 n0 = K.tlas_nodes[tcur * 4 + 0], n1 = K.tlas_nodes[tcur * 4 + 1], n2 = K.tlas_nodes[tcur * 4 + 2],
                                 n3 = K.tlas_nodes[tcur * 4 + 3];
                    float tL, tR;
                    const bool hL = slab(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), w, tMax, tL);
                    const bool hR = slab(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), w, tMax, tR);
                    const int Lr = __float_as_int(n3.x), Rr = __float_as_int(n3.y);
                    if (hL && hR) {
                        L.tstack[tsp * 64 + lane] = make_uint2((uint32_t)Rr, 0u);
                        ++tsp;
                        tcur = Lr;
                    } else if (hL || hR) {
                        tcur = hL ? Lr : Rr;
                    } else {
                        pop_t();
                    }
                } else {
                    lf = K.tlas_leaves[~tcur];
                    has = true;
                }
            }
            need_pop = has;
            TS_ADDQ(9, t_walk);
            if (has)
                L.leafx[step * 64 + lane] = lf.x; // (every lane uses slot `step` here: no per-slot results to keep apart)
            if (!__builtin_amdgcn_ballot_w64(has))
                break;
            any_leaf = true;
            const unsigned long long t_b = TS_NOW();
            base = build_pairs_general<true>(K, L, lane, has, lf, w, tMax, step, base);
            TS_ADDQ(10, t_b);
            if (base >= TLAS_FILL_TARGET)
                break;
        }
        if (!any_leaf)
            break;
        wave_sync();
        const unsigned long long t_q = TS_NOW();
        run_any_queue<1>(K, L, lane, base, o, d);
        wave_sync();
        TS_ADDQ(8, t_q);
#ifdef PT_TRAV_STATS
        cyc.c[7] += 1; // fills (any)
#endif
        if (L.occ[lane] != 0u)
            t_active = need_pop = false; // blocked: nothing more to look for
        wave_sync();
    }
    return alive && (L.occ[lane] != 0u);
}

// Shadow rays: the any-hit walk (bvh_any_hit_tlas, intersection.cuh:481-524).  The answer is an OR over every leaf the walk
// reaches, in any order and with nothing to cull by, so a fill does not stop at one leaf per ray: every lane keeps walking
// the TLAS, leaf after leaf, and the wave keeps appending (ray, mesh) pairs -- the entry carries the TLAS index itself, so
// no per-slot bookkeeping -- until no lane has a leaf left or the list cannot take the next leaf's hits; THEN the queue
// runs, once, over all of them (one leaf per round was 1.5 pairs per queue run on the 136-mesh scene: 2.4 M runs per frame,
// 41 % of a wave's life).  A leaf's hits are never split between fills (they are counted before they are appended); a
// ray seen blocked drops its remaining pairs at refill and stops walking at the next fill.
PT_DEV bool any_hit_pairs_tlas(const KParams &K, const PairLds &L, int lane, bool alive, f3 o, f3 d, float tMax, CycleAcc &cyc) {
    if (K.tlas_any_rounds)
        return any_hit_pairs_tlas_rounds(K, L, lane, alive, o, d, tMax, cyc);
    const RayO w = make_ray(o, d);
    float tE;
    bool t_active = alive && slab(tlas_bmin(K), tlas_bmax(K), w, tMax, tE);
    ((float *)L.best)[lane] = tMax;
    L.occ[lane] = 0u;
    int tcur = K.tlas_root_ref, tsp = 0;
    bool need_pop = false;
    auto pop_t = [&]() {
        t_active = false;
        if (tsp > 0) {
            --tsp;
            tcur = (int)L.tstack[tsp * 64 + lane].x;
            t_active = true;
        }
    };
    const int cap = K.tlas_max_leaf * 64 + TLAS_FILL_TARGET; // entries of the pair list (carve_pair_lds)
    uint32_t held = 0u; // a leaf's hits that did not fit the previous fill
    int held_first = 0;
    bool holding = false; // (wave-uniform)
    wave_sync();
    for (;;) {
        int base = 0;
        for (;;) {
            uint32_t mask = held;
            int first = held_first;
            if (!holding) {
                int2 lf = make_int2(0, 0);
                bool has = false;
                const unsigned long long t_walk = TS_NOW();
                if (need_pop)
                    pop_t();
                while (t_active && !has) {
                    if (tcur >= 0) {
                        const float4 n0 = K.tlas_nodes[tcur * 4 + 0], n1 = K.tlas_nodes[tcur * 4 + 1], n2 = K.tlas_nodes[tcur * 4 + 2],
                                     n3 = K.tlas_nodes[tcur * 4 + 3];
                        float tL, tR;
                        const bool hL = slab(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), w, tMax, tL);
                        const bool hR = slab(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), w, tMax, tR);
                        const int Lr = __float_as_int(n3.x), Rr = __float_as_int(n3.y);
                        if (hL && hR) {
                            L.tstack[tsp * 64 + lane] = make_uint2((uint32_t)Rr, 0u);
                            ++tsp;
                            tcur = Lr;
                        } else if (hL || hR) {
                            tcur = hL ? Lr : Rr;
                        } else {
                            pop_t();
                        }
                    } else {
                        lf = K.tlas_leaves[~tcur];
                        has = true;
                    }
                }
                need_pop = has;
                TS_ADDQ(9, t_walk);
                if (!__builtin_amdgcn_ballot_w64(has))
                    break;
                const unsigned long long t_b = TS_NOW();
                mask = root_masks<true>(K, lane, has, lf, w, tMax);
                first = lf.x;
                TS_ADDQ(10, t_b);
            }
            // hits of this leaf over the wave (a lane has at most 32): bit planes of the lanes' counts
            int total = 0;
            const int cnt = __builtin_popcount(mask);
#pragma unroll
            for (int b = 0; b < 6; ++b)
                total += __builtin_popcountll(__builtin_amdgcn_ballot_w64((cnt >> b) & 1)) << b;
            if (base + total > cap) { // (never with base == 0: one leaf's hits always fit)
                held = mask;
                held_first = first;
                holding = true;
                break;
            }
            holding = false;
            held = 0u;
            base = append_pairs(L, lane, mask, base, [first](int i) { return first + i; });
        }
        if (base == 0)
            break; // (no lane found another leaf, and nothing is held back)
        wave_sync();
        const unsigned long long t_q = TS_NOW();
        run_any_queue<2>(K, L, lane, base, o, d);
        wave_sync();
        TS_ADDQ(8, t_q);
#ifdef PT_TRAV_STATS
        cyc.c[7] += 1; // fills (any)
#endif
        if (L.occ[lane] != 0u) {
            t_active = need_pop = false; // blocked: nothing more to look for
            held = 0u;
        }
        wave_sync();
    }
    return alive && (L.occ[lane] != 0u);
}

} // namespace pt
#include "pt_merged.hip.h"
namespace pt {

// ---------------------------------------------------------------------------------
#ifndef PT_WAVES_PER_EU
#define PT_WAVES_PER_EU 3
#endif
// PMODE 1: the phases of the render loop that run at s_setprio 1 -- bit 0 [R]+[A], 1 [B], 2 [C], 3 [C2], 4 [D], 5 [E] -- the others
// at 0.  The traversal phases are short chains of LDS reads, each waited for: a wave in one of them that loses the issue
// arbitration to four waves in their VALU-dense shading phases leaves its LDS requests unissued.
#ifndef PT_PRIO_MASK
#define PT_PRIO_MASK 0x1b
#endif
template <int PMODE, int PHASE, int PREV> PT_DEV void phase_prio() {
    if (PMODE == 1 && ((PT_PRIO_MASK >> PHASE) & 1) != ((PT_PRIO_MASK >> PREV) & 1))
        __builtin_amdgcn_s_setprio((PT_PRIO_MASK >> PHASE) & 1);
}

// Waves per SIMD a variant is built for (= its register budget: 512 / waves, in steps of 8).  PMODE 1 with the simple
// materials -- a whole small scene in 7.5 KB of LDS, no traversal stacks -- runs five (96 VGPRs; the Cornell kernel then
// spills 22 registers, 88 B per lane, and is still 4 % faster: 2.233 -> 2.150 ms).  Its FULL variant would spill 168 B per
// lane (measured on Cornell with force_full: 2.29 -> 2.69 ms) and stays at four; the modes with per-lane stacks need their
// 10 KB of LDS per wave, which caps a CU at 16 waves whatever the registers (and at 96 VGPRs they spill 83-120 registers).
#ifndef PT_WAVES_PMODE1
#define PT_WAVES_PMODE1 5
#endif
// Two tiles per workgroup (WG = 2) halve what a wave costs in LDS for the shared copies: the small scene then fits SIX waves
// per SIMD (80 VGPRs, 16 of them spilled) -- 24 waves per CU instead of 20.
#ifndef PT_WAVES_PMODE1_WG2
#define PT_WAVES_PMODE1_WG2 6
#endif
constexpr int waves_per_simd(int pmode, bool full, int wg = 1) {
    return (pmode == 1 && !full) ? (wg == 2 ? PT_WAVES_PMODE1_WG2 : PT_WAVES_PMODE1) : PT_WAVES_PER_EU;
}
// LDS a one-wave workgroup may use without lowering that occupancy: a CU has 160 KB, allocated in 1280-byte granules
// (measured: Cornell at 7,680 B runs 20 waves per CU, at 7,744 B visibly fewer; showcase at 10,192 B 16, at 10,384 B fewer)
constexpr int LDS_GRANULE = 1280;
constexpr int lds_per_wave(int pmode, bool full, int wg = 1) { return 160 * 1024 / (4 * waves_per_simd(pmode, full, wg)) / LDS_GRANULE * LDS_GRANULE; }
// (a workgroup of wg waves may use wg times that: the granule rounding applies to the workgroup's total)
constexpr int lds_per_workgroup(int pmode, bool full, int wg) { return 160 * 1024 / (4 * waves_per_simd(pmode, full, wg) / wg) / LDS_GRANULE * LDS_GRANULE; }
// PMODE 0: lock-step mesh loop; 1: pair compaction, single-leaf BLASes (triangles staged in LDS);
//       2: pair compaction, general BLASes (per-lane traversal, LDS stacks); 3: the same behind a real TLAS, in rounds;
//       4: as 2 with ONE traversal per iteration: a light sample's shadow ray rides with the next extension ray
//          (pt_merged.hip.h)
//  WG: waves per workgroup.  1 = one 8x8 tile per workgroup.  4 (PMODE 2 only, option lds_nodes): four tiles per
//      workgroup that share one LDS copy of the mesh heads and of the top TOP_LEVELS levels of every BLAS.
typedef const __attribute__((address_space(4))) KParams *kparams_ptr;
PT_DEV const KParams &kparams(kparams_ptr p) {
    asm volatile("" : "+s"(p));
    return *(const KParams *)p;
}
template <int GEOM, bool FULL, int PMODE, int WG = 1, bool STREAM = false>
__global__ __launch_bounds__(64 * WG) __attribute__((amdgpu_waves_per_eu(waves_per_simd(PMODE, FULL, WG), 8))) void path_trace_kernel(const KParams Kin) {
    // The parameters are read where they are used, through the kernarg segment itself (scalar loads that hit the constant
    // cache), not out of the by-value copy: ~110 dwords of pointers, camera and options held in SGPRs across the persistent
    // loop were what the kernel spilled -- 93 scalar registers parked in VGPR lanes and ~115 v_readlane per iteration.  Each
    // phase gets its own OPAQUE copy of the pointer (kparams: an empty asm the compiler cannot see through), so a field's
    // live range ends with the phase that reads it.
    (void)Kin;
    const kparams_ptr kp0 = (kparams_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    const KParams &K = kparams(kp0); // set-up and epilogue
    extern __shared__ uint2 lds_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LdsStack stk{lds_raw + lane};
    PairLds PL{};
    constexpr bool MERGED = (PMODE == 4);
    if (PMODE == 1) {
        PL = carve_pm1((void *)lds_raw, K, wave);
        const int2 lf = K.tlas_leaves[~K.tlas_root_ref];
        // mesh i's packets start PAIR_PAD * i float4 later than in the arena: with 48-B packets the per-mesh blocks would
        // otherwise sit a multiple of 16 banks apart and lanes working on different meshes would collide on ds_read_b128
        for (int i = 0; i < K.pair_meshes; ++i) {
            const int m = K.tlas_mesh_ids[lf.x + i];
            const int2 leaf = K.leaves[~__float_as_int(K.mesh_recs[m * MESH_REC_F4].w)];
            for (int k = threadIdx.x; k < leaf.y * 3; k += 64 * WG)
                PL.tris[leaf.x * 3 + i * PAIR_PAD + k] = K.tris[leaf.x * 3 + k];
        }
        for (int i = threadIdx.x; i < K.pair_meshes; i += 64 * WG) {
            const int m = K.tlas_mesh_ids[lf.x + i];
            const MeshHead mh = load_mesh_head(K, m);
            PL.meshbox[2 * i] = K.mesh_recs[m * MESH_REC_F4 + 0];
            float4 hb = K.mesh_recs[m * MESH_REC_F4 + 1];
            hb.w = __int_as_float((mh.flags & 0xff) | (m << 8)); // (flags and mesh id in one word: staged_mesh_head)
            PL.meshbox[2 * i + 1] = hb;
            const int2 leaf = K.leaves[~mh.root_ref];
            PL.meshtab[i] = make_int4(leaf.x, leaf.y, mh.flags, m);
        }
        // (the barrier follows the shading inputs, below)
    } else if (WG > 1) {
        PL = carve_pair_lds_wg((void *)lds_raw, wave, K.pair_meshes, K.stack_entries);
        const int2 lf = K.tlas_leaves[~K.tlas_root_ref];
        for (int i = threadIdx.x; i < K.pair_meshes; i += 64 * WG) {
            const int m = K.tlas_mesh_ids[lf.x + i];
            const MeshHead mh = load_mesh_head(K, m);
            PL.meshbox[2 * i] = K.mesh_recs[m * MESH_REC_F4 + 0];
            float4 hb = K.mesh_recs[m * MESH_REC_F4 + 1];
            hb.w = __int_as_float((mh.flags & 0xff) | (m << 8));
            PL.meshbox[2 * i + 1] = hb;
        }
        // the first TOP_NODES nodes of each tree (its top TOP_LEVELS levels: the host numbers them in level order)
        float4 *top = const_cast<float4 *>(PL.topnodes);
        for (int i = threadIdx.x; i < K.pair_meshes * TOP_NODES * 4; i += 64 * WG) {
            const int o = i / (TOP_NODES * 4), k = (i / 4) % TOP_NODES;
            const int root = __float_as_int(K.mesh_recs[K.tlas_mesh_ids[lf.x + o] * MESH_REC_F4].w);
            int node = (root >= 0 ? root : 0) + k;
            node = node < K.n_nodes ? node : (K.n_nodes > 0 ? K.n_nodes - 1 : 0);
            top[i] = K.n_nodes > 0 ? K.nodes[node * 4 + (i & 3)] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        __syncthreads();
        if (K.top_off)
            PL.topnodes = nullptr;
    } else if (PMODE) {
        PL = carve_pair_lds((void *)lds_raw, 0, PMODE == 3 ? 0 : K.pair_meshes, K.stack_entries,
                            PMODE == 3 ? K.tlas_max_leaf : 0, PMODE == 3 ? K.tlas_depth : 0, MERGED ? K.pair_cap : 0);
        const int2 lf = PMODE == 3 ? make_int2(0, 0) : K.tlas_leaves[~K.tlas_root_ref];
        for (int i = lane; PMODE != 3 && i < K.pair_meshes; i += 64) {
            const int m = K.tlas_mesh_ids[lf.x + i];
            const MeshHead mh = load_mesh_head(K, m);
            PL.meshbox[2 * i] = K.mesh_recs[m * MESH_REC_F4 + 0];
            float4 hb = K.mesh_recs[m * MESH_REC_F4 + 1];
            hb.w = __int_as_float((mh.flags & 0xff) | (m << 8)); // (flags and mesh id in one word: staged_mesh_head)
            PL.meshbox[2 * i + 1] = hb;
        }
        __syncthreads();
    }
    // STREAM (one-wave workgroups; instantiated for PMODE 1): the launch is a grid of PERSISTENT waves and a lane that has
    // finished its pixel takes the next one of the launch -- [R] in the loop below -- instead of idling until the slowest
    // pixel of its 8x8 tile is done (tools/trav_stats.py: 18 % of the lane-iterations of a 1080p Cornell frame).
    static_assert(!STREAM || WG == 1, "lane refill: one-wave workgroups");
    // K.tile_run (one-tile workgroups): XCD-aware order.  The dispatcher deals consecutive workgroups to the eight XCDs in turn, so
    // by number each XCD -- each with an L2 of its own -- renders every eighth tile of a row and all eight fetch the same nodes
    // and triangles.  Instead, of every 8 * run consecutive tiles XCD x takes tiles [x * run, (x + 1) * run): neighbours, whose
    // rays walk the same part of the trees, share an L2.  Runs of 8 (64 x 8 pixels): fluid frame 0.888 -> 0.860 ms alone, 0.813 ->
    // 0.791 overlapping, with its refit 0.941 -> 0.913; showcase 3.45 -> 3.42 / 3.18 -> 3.14, at 4K 24.5 -> 24.1; 4 and 16 are within
    // half a percent of 8, 2 gains half as much; blocks of 4 x 2 .. 8 x 8 tiles per XCD instead of runs lose 1 .. 9 %, whole
    // vertical stripes per XCD 10 % (the XCDs' shares of the work differ).  What else was measured about the ORDER of the tiles, all slower than by number, because
    // the ~5,000 tiles in flight stop being neighbours: bottom-up +13 % (showcase) / +28 % (fluid), every tile at random +32 / +42 %,
    // blocks of 32 at random +26 / +39 %, rows from top and bottom in turn +25 / +47 %, rows with a stride of 4 / 8 / 16 +23..33 %
    // / +34..43 %, Z-order +4 / +16 %; and the dearest tiles of an earlier frame first (timed per tile, counting-sorted into 256
    // classes on the device): -3 % for a frame alone on the chip, whose tail it shortens, +4 % for overlapping fluid frames --
    // per 32-tile block instead of per tile +15 %: a launch whose resident waves all start in step stays in step.
    // Non-temporal (`nt`) loads and stores for the per-pixel streams -- generator states, HDR image, G-buffers, RGB8 --, so that they
    // would not displace the trees in the L2: showcase -0.4 %, fluid +0.9 %, 1 M triangles +0.7..2 % (the states of a 1 ms frame are
    // still in the 256 MB last-level cache when the next frame wants them): dropped.
    // (A trailing run of fewer than 8 * run tiles keeps its numbers.)
    int tile_sel = (int)blockIdx.x;
    if (!STREAM && WG == 1 && K.tile_run > 0) {
        const int span = 8 * K.tile_run, chunk = (int)blockIdx.x / span, in = (int)blockIdx.x % span;
        if ((chunk + 1) * span <= (int)gridDim.x)
            tile_sel = chunk * span + (in % 8) * K.tile_run + in / 8;
    }
    const int tile = WG > 1 ? blockIdx.x * WG + wave : tile_sel;
    // (a larger workgroup's last tiles may not exist: such a wave takes part in the staging and the barrier with a tile
    // outside the frame -- `inside` is false for all its lanes -- and leaves before the loop)
    const bool no_tile = WG > 1 && tile >= K.n_tiles;
    if (PMODE != 1 && no_tile)
        return; // (after the workgroup's only barrier)
    // (a frame dealt to several concurrent launches, ptrt_set_option "split": this launch's k-th row of tiles is row
    // split_i + k * split_n of the frame; the counters' slot follows the frame's numbering)
    const int tx = no_tile ? 0 : tile % K.tiles_x;
    const int ty = no_tile ? 0 : (K.split_n > 1 ? (tile / K.tiles_x) * K.split_n + K.split_i : tile / K.tiles_x);
    // Registers are what this kernel runs out of (128 per lane at four waves per SIMD; what does not fit is spilled to
    // scratch, and a reload is a trip to the L2).  State that is only touched when a path starts or ends stays out of
    // them: the pixel's coordinates are recomputed from the lane id where they are needed (the empty asm keeps the
    // compiler from hoisting them back into the loop's live set), the ray counters are wave totals in scalar registers.
    // (The running sum of the samples was tried in the pixel's own accum words, read-modify-write at the end of a
    // sample: three registers fewer, but a dependent global load in most iterations -- Cornell 2.40 -> 2.54 ms.)
    int pxy = -1; // STREAM: the lane's pixel, x | local row << 16; -1: none
    auto px = [&]() {
        if (STREAM)
            return pxy & 0xffff;
        int l = lane;
        asm volatile("" : "+v"(l));
        return tx * 8 + (l & 7);
    };
    auto pyl = [&]() {
        if (STREAM)
            return pxy >> 16;
        int l = lane;
        asm volatile("" : "+v"(l));
        return ty * 8 + (l >> 3);
    };
    auto pidx = [&](int width) { return (size_t)pyl() * width + px(); };
    const bool inside = !STREAM && (px() < K.width) && (pyl() < K.rows);
    // PMODE 1 (a scene small enough for its triangles to sit in LDS): what the shading phases would otherwise fetch from
    // global memory in every iteration is staged too -- the jitter table and the lane's blue-noise value for [A], the light
    // records for [C], the material records by mesh order; [A], [C], [C2] and [E] then read no global memory.  Worth 2 %
    // (Cornell 2.27 -> 2.23 ms), and only while the workgroup's LDS stays within 10 KB: the kernel runs 16 waves per CU on
    // registers, one wave less costs 6.5 % -- which is why the modes whose stacks fill that budget do not stage (measured:
    // showcase +9 % with 64 bytes too many).
    constexpr bool STAGED = (PMODE == 1);
    bool lights_lds = false, mats_lds = false, jit_lds = false, bn_lds = false;
    constexpr int MAT_F4 = FULL ? 6 : 3; // float4 per staged material record (the simple-material kernel reads the first three)
    if (STAGED) {
        if (K.lds_flags) { // (0: the launch has no room for them)
            float2 *jit = const_cast<float2 *>(PL.jit), *bn = const_cast<float2 *>(PL.bn);
            float4 *lights = const_cast<float4 *>(PL.lights), *mats = const_cast<float4 *>(PL.mats);
            if (threadIdx.x < 16)
                jit[threadIdx.x] = taa_table_entry(threadIdx.x);
            bn_lds = (K.lds_flags & 8) != 0;
            if (!STREAM && bn_lds) { // (STREAM: a lane's entry follows its pixel, [R])
                const int x0 = px(), y0 = global_row(pyl(), K.y0, K.il_period, K.il_phase);
                bn[lane] = K.blue_noise[(y0 & 63) * 64 + (x0 & 63)];
            }
            lights_lds = (K.lds_flags & 1) != 0;
            jit_lds = (K.lds_flags & 4) != 0;
            mats_lds = (K.lds_flags & 2) != 0;
            if (lights_lds)
                for (int i = threadIdx.x; i < K.n_lights * 4; i += 64 * WG)
                    lights[i] = K.lights[i];
            if (mats_lds) {
                const int2 lf = K.tlas_leaves[~K.tlas_root_ref];
                for (int i = threadIdx.x; i < K.pair_meshes * MAT_F4; i += 64 * WG)
                    mats[i] = K.materials[K.tlas_mesh_ids[lf.x + i / MAT_F4] * 6 + i % MAT_F4];
            }
        }
        __syncthreads(); // the workgroup's only barrier: triangles, tables and shading inputs are in place
        if (no_tile)
            return;
    }

    Rng rng = {0, 0, 0, 0, 0, 0};
    if (inside) {
        const size_t idx = pidx(K.width), npix = K.rng_plane;
        rng.d = K.rng[idx];
        rng.v0 = K.rng[npix + idx];
        rng.v1 = K.rng[2 * npix + idx];
        rng.v2 = K.rng[3 * npix + idx];
        rng.v3 = K.rng[4 * npix + idx];
        rng.v4 = K.rng[5 * npix + idx];
    }
    f3 avg_color = mk3(0.0f);
    auto close_sample = [&](f3 a) { avg_color = avg_color + a; }; // scene_kernels.cuh:171-176
    uint32_t n_ext = 0, n_shadow = 0, n_zero = 0; // wave totals (uniform)
    // The pair modes have no scalar registers to spare: two loop-carried counters ended up in a VGPR lane that itself lived in
    // scratch -- a load, a v_writelane and a store per iteration (0.5 GB of writes per 1080p Cornell frame at five waves per
    // SIMD).  The totals sit in LDS instead, one 64-bit add per iteration: {extension rays, shadow rays << 32}.
    constexpr bool LDS_COUNT = (PMODE >= 1) && (WG == 1 || PMODE == 1);
    unsigned long long *lds_count = LDS_COUNT ? PL.count : nullptr;
    if (LDS_COUNT) {
        if (lane == 0)
            lds_count[0] = lds_count[1] = 0ull;
        wave_sync();
    }

    // sample index (low half) and bounce (high half) of the lane in ONE register (ptrt_render refuses spp or max_depth beyond 32767)
    int sb = inside ? 0 : K.spp; // (STREAM: no lane has a pixel yet)
    bool fresh = true;
    f3 ro = mk3(0.0f), rd = mk3(0.0f);
    bool ray_spec = true, prev_was_specular = true;
    f3 throughput = mk3(1.0f), acc = mk3(0.0f);
    // PMODE 4: the light sample parked with its shadow ray until the next traversal has answered its visibility
    bool pending = false, fin = false;
    f3 pend = mk3(0.0f), park_o = mk3(0.0f), park_d = mk3(0.0f, 0.0f, 1.0f);
    float park_tmax = 0.0f;

    CycleAcc cyc;
    PL.cyc = &cyc;
    const unsigned long long t_kernel = TS_NOW();
    // STREAM: the launch's tiles are numbered 0 .. n_tiles - 1 and handed out by tickets (K.queue[0]; one atomic per tile, the
    // next ticket is drawn when the current tile is opened, so its latency hides behind that tile's work).  Wave-uniform:
    // (A wave's FIRST tile is its own number -- no ticket, no burst of thousands of atomics on one address before anything
    // runs; ticket k is tile gridDim.x + k.)
    int q_org = 0, q_next = 64;    // the tile being handed out (x0 | local y0 << 16) and the next of its 64 pixels
    uint32_t q_ticket = blockIdx.x; // lane 0: the tile drawn ahead
    bool q_first = true;
    int q_sub = 0;        // tiles left on the current ticket
    uint32_t q_tile = 0u; // the open tile's number
    bool q_open = STREAM;
    uint32_t n_px = 0;             // pixels this wave took (counters)
    for (;;) {
        phase_prio<PMODE, 0, 5>();
        if (STREAM) {
            // ---- [R] lanes without work: write the finished pixel out, take the next one.  Which lane renders a pixel
            // changes nothing in it: generator state, samples and sums are the pixel's own.
            const KParams &KR = kparams(kp0);
            const unsigned long long t_r = TS_NOW();
            const bool idle = ((sb & 0xffff) >= KR.spp) && !(MERGED && pending);
            if (__builtin_amdgcn_ballot_w64(idle && (pxy >= 0 || q_open))) {
                // Order matters for what the wave waits for: the new pixel is chosen FIRST and its blue-noise entry requested
                // before anything else is in flight, so that the one wait of this block (the entry goes into the lane's LDS
                // slot, [A] reads it at once) is for that load alone -- not for the nine stores of the finished pixel and
                // the six loads of the new generator state behind it, which nothing needs before [C].
                const int old = idle ? pxy : -1;
                int got = -1;
                while (q_open) {
                    const bool wants = idle && got < 0;
                    const unsigned long long want = __builtin_amdgcn_ballot_w64(wants);
                    if (!want)
                        break;
                    if (q_next == 64) { // open the next tile
                        // (a ticket is good for KR.ticket_tiles consecutive tiles)
                        uint32_t t;
                        if (q_sub > 0) {
                            t = q_tile + 1u;
                            --q_sub;
                        } else {
                            t = ((uint32_t)__builtin_amdgcn_readfirstlane((int)q_ticket) + (q_first ? 0u : gridDim.x)) * (uint32_t)KR.ticket_tiles;
                            q_first = false;
                            q_sub = KR.ticket_tiles - 1;
                            if (t < (uint32_t)KR.n_tiles && lane == 0) // (-amdgpu-atomic-optimizer-strategy=None: the result is not needed before the next ticket's first tile)
                                q_ticket = atomicAdd(KR.queue, 1u);
                        }
                        if (t >= (uint32_t)KR.n_tiles) {
                            q_open = false;
                            break;
                        }
                        q_tile = t;
                        const int trow = (int)t / KR.tiles_x, tcol = (int)t - trow * KR.tiles_x;
                        q_org = (tcol * 8) | ((KR.split_n > 1 ? trow * KR.split_n + KR.split_i : trow) * 8) << 16;
                        q_next = 0;
                    }
                    const int rank = lane_prefix(want), avail = 64 - q_next, asked = __builtin_popcountll(want);
                    if (wants && rank < avail) {
                        const int pix = q_next + rank;
                        const int x = (q_org & 0xffff) + (pix & 7), yl = (q_org >> 16) + (pix >> 3);
                        if (x < KR.width && yl < KR.rows) { // (else: the lane asks again in the next round)
                            got = x | (yl << 16);
                        }
                    }
                    n_px += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(wants && got >= 0));
                    q_next += asked < avail ? asked : avail;
                }
                // (the finished pixel's state is stored LAST -- nothing waits for stores then --, so it has to outlive the loads
                // of the new pixel's: the six generator words wait in the wave's traversal scratch in LDS (minima + pair list,
                // idle between [D] and [B]; plane k at word 64 k + lane), the radiance sum in three registers.  Nine spare
                // registers at the one point of the loop where every lane's whole path state is live were what this kernel spilled.)
                uint32_t *park = (uint32_t *)PL.best;
                if (old >= 0) {
                    park[lane] = rng.d;
                    park[64 + lane] = rng.v0;
                    park[128 + lane] = rng.v1;
                    park[192 + lane] = rng.v2;
                    park[256 + lane] = rng.v3;
                    park[320 + lane] = rng.v4;
                }
                const f3 avg_old = avg_color;
                TS_LANES(22, got >= 0);
                TS_LANES(23, true);
                if (got >= 0) {
                    float2 bnv = make_float2(0.0f, 0.0f);
                    if (STAGED) { // (no memory operation inside the loop above, none conditional here: the compiler's
                                  // wait-count pass is flow-insensitive and would wait for everything in flight)
                        const int gy = global_row(got >> 16, KR.y0, KR.il_period, KR.il_phase);
                        bnv = KR.blue_noise[(gy & 63) * 64 + (got & 63)];
                    }
                    // (one 32-bit pixel index against six scalar plane bases: no 64-bit address arithmetic per plane)
                    const uint32_t idx = (uint32_t)(got >> 16) * (uint32_t)KR.width + (uint32_t)(got & 0xffff);
                    __builtin_assume(idx < (1u << 27));
                    const size_t npix = KR.rng_plane;
                    rng.d = (KR.rng)[idx];
                    rng.v0 = (KR.rng + npix)[idx];
                    rng.v1 = (KR.rng + 2 * npix)[idx];
                    rng.v2 = (KR.rng + 3 * npix)[idx];
                    rng.v3 = (KR.rng + 4 * npix)[idx];
                    rng.v4 = (KR.rng + 5 * npix)[idx];
                    if (STAGED) {
                        asm volatile("" ::"v"(bnv.x), "v"(bnv.y)); // (the entry has arrived, whichever way the next branch goes)
                        if (bn_lds)
                            const_cast<float2 *>(PL.bn)[lane] = bnv;
                    }
                    pxy = got;
                    avg_color = mk3(0.0f);
                    sb = 0;
                    fresh = true;
                } else if (old >= 0) {
                    pxy = -1;
                }
                if (old >= 0) {
                    const uint32_t idx = (uint32_t)(old >> 16) * (uint32_t)KR.width + (uint32_t)(old & 0xffff);
                    __builtin_assume(idx < (1u << 27));
                    const size_t npix = KR.rng_plane;
                    (KR.rng)[idx] = park[lane];
                    (KR.rng + npix)[idx] = park[64 + lane];
                    (KR.rng + 2 * npix)[idx] = park[128 + lane];
                    (KR.rng + 3 * npix)[idx] = park[192 + lane];
                    (KR.rng + 4 * npix)[idx] = park[256 + lane];
                    (KR.rng + 5 * npix)[idx] = park[320 + lane];
                    // (a power-of-two sample count divides exactly by multiplication: the same bits as the division)
                    const float n = (float)KR.spp;
                    const f3 out = (KR.spp & (KR.spp - 1)) == 0 ? avg_old * (1.0f / n) : avg_old / n;
                    const uint32_t i3 = idx * 3u;
                    KR.accum[i3 + 0] = out.x;
                    KR.accum[i3 + 1] = out.y;
                    KR.accum[i3 + 2] = out.z;
                }
            }
            if (PMODE == 1)
                TS_ADD(15, t_r);
        }
        const KParams &KL = kparams(kp0);
        if (!__builtin_amdgcn_ballot_w64((sb & 0xffff) < KL.spp || (MERGED && pending)))
            break;
        const bool live = (sb & 0xffff) < KL.spp;
#ifdef PT_TRAV_STATS
        {
            const unsigned long long lm = __builtin_amdgcn_ballot_w64(live);
            if (lane == 0) {
                atomicAdd(&g_trav_stats[16], 1ull);
                atomicAdd(&g_trav_stats[17], (unsigned long long)__builtin_popcountll(lm));
            }
        }
#endif
        PT_MARK("A");
        const KParams &KA = kparams(kp0);
        const unsigned long long t_pa = TS_NOW();
        // ---- [A] primary ray (scene_kernels.cuh:147-167, camera.cuh:156-205)
        // K.sample_sync: a lane whose path has ended starts its next sample only when NO lane of the wave is in the middle of one.
        // The lanes then sit at the same bounce: a first hit samples no light (ray_spec), so the light-sample phases [C2] and
        // [D] are skipped by the whole wave in that iteration instead of running for the half of the lanes that are deeper,
        // and [A] runs once per sample for 64 lanes instead of every iteration for a quarter of them.
        const bool hold = KA.sample_sync && __builtin_amdgcn_ballot_w64(live && !fresh) != 0ull;
        TS_LANES(18, live && fresh && !hold);
        if (live && fresh && !hold) {
            const int x = px(), y = global_row(pyl(), KA.y0, KA.il_period, KA.il_phase);
            float tjx, tjy, bnx, bny;
            if (STAGED && jit_lds) {
                const float2 e = lds_ld2(PL.jit, (KA.frame_count + (sb & 0xffff)) % 16);
                tjx = e.x;
                tjy = e.y;
                if (bn_lds) {
                    int l = lane;
                    asm volatile("" : "+v"(l));
                    blue_noise_shift(lds_ld2(PL.bn, l), KA.frame_count + (sb & 0xffff), bnx, bny);
                } else { // (samples in step: [A] runs once per sample for the whole wave -- the entry straight from the table)
                    blue_noise_jitter(KA.blue_noise, x, y, KA.frame_count + (sb & 0xffff), bnx, bny);
                }
            } else {
                taa_jitter(KA.frame_count + (sb & 0xffff), tjx, tjy);
                blue_noise_jitter(KA.blue_noise, x, y, KA.frame_count + (sb & 0xffff), bnx, bny);
            }
            const float jitter_x = tjx + (bnx - 0.5f) * 0.25f;
            const float jitter_y = tjy + (bny - 0.5f) * 0.25f;
            const float u = ((float)x + 0.5f + jitter_x) / (float)KA.width;
            const float v = 1.0f - ((float)y + 0.5f + jitter_y) / (float)KA.height;
            if (KA.cam.lens_radius <= 0) {
                const f3 dir = KA.cam.llc + u * KA.cam.horizontal + v * KA.cam.vertical - KA.cam.origin;
                ro = KA.cam.origin;
                rd = normalize(dir);
            } else {
                f3 p;
                do {
                    const float a = rng_uniform(rng);
                    const float b = rng_uniform(rng);
                    p = 2.0f * mk3(a, b, 0.0f) - mk3(1.0f, 1.0f, 0.0f);
                } while (dot(p, p) >= 1.0f);
                const f3 rdisk = KA.cam.lens_radius * p;
                const f3 offset = KA.cam.u * rdisk.x + KA.cam.v * rdisk.y;
                const f3 dir = KA.cam.llc + u * KA.cam.horizontal + v * KA.cam.vertical - KA.cam.origin - offset;
                ro = KA.cam.origin + offset;
                rd = normalize(dir);
            }
            ray_spec = true;
            prev_was_specular = true;
            throughput = mk3(1.0f);
            if (!MERGED) // (PMODE 4 clears `acc` when it closes a sample: the previous one may still be open here)
                acc = mk3(0.0f);
            sb &= 0xffff; // bounce = 0
            fresh = false;
        }

        if (PMODE == 1 || TS_SHADING)
            TS_ADD(8, t_pa);
        const bool act = live && !fresh; // (has a ray: every live lane, unless K.sample_sync keeps it waiting)
#ifdef PT_TRAV_STATS
        {
            const unsigned long long am = __builtin_amdgcn_ballot_w64(act);
            PL.stat_bounce = am ? __builtin_amdgcn_readlane(sb >> 16, __builtin_ctzll(am)) : -1;
        }
#endif
        PT_MARK("B");
        phase_prio<PMODE, 1, 0>();
        const KParams &KB = kparams(kp0);
        // ---- [B] closest hit, all live lanes together (PMODE 4: and the parked shadow rays in the same traversal)
        Hit h;
        int h_order = 0; // PMODE 1: the hit mesh's place in the leaf (what the staged tables are indexed by)
        if (MERGED) {
            bool blocked = false;
            const unsigned long long t_tr = TS_NOW();
            trace_merged(KB, PL, lane, act, ro, rd, pending, park_o, park_d, park_tmax, h, blocked, cyc);
            TS_ADD(12, t_tr);
            if (pending && !blocked)
                acc = acc + pend; // the light sample of the previous vertex (path_logic.cuh:840-867), in its place
            pending = false;
            if (fin) { // that vertex was the last of its path
                acc = clamp_vector_soft(acc, 100.0f);
                close_sample(acc);
                acc = mk3(0.0f);
                fin = false;
            }
        } else {
            const unsigned long long t_tr = TS_NOW();
            h = (PMODE == 1)   ? closest_hit_pairs(KB, PL, lane, act, ro, rd, h_order)
                : (PMODE == 2) ? closest_hit_pairs_dyn(KB, PL, lane, act, ro, rd)
                : (PMODE == 3) ? closest_hit_pairs_tlas(KB, PL, lane, act, ro, rd, cyc)
                               : closest_hit<GEOM>(KB, act, ro, rd, stk);
            TS_ADD(12, t_tr);
        }

        PT_MARK("C");
        phase_prio<PMODE, 2, 1>();
        const KParams &KC = kparams(kp0);
        const unsigned long long t_pc = TS_NOW();
        // ---- [C] first half of the shading
        bool end_path = false, shaded = false, want_shadow = false;
        Surface hit;
        hit.point = hit.normal = mk3(0.0f);
        hit.t = 0.0f;
        hit.front_face = true;
        f3 L = mk3(0.0f), light_scale = mk3(0.0f), shadow_o = mk3(0.0f);
        float pdf_sample = 1.0f, shadow_tmax = 0.0f, light_att = 1.0f;
        if (!LDS_COUNT)
            n_ext += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act));
        if (act) {
            if (h.mesh < 0) {
                if (sb == 0) { // G-buffer of the first sample's first hit (scene_kernels.cuh:181-193)
                    const size_t idx = pidx(KC.width);
                    float zero = 0.0f; // (opaque: hoisted out of the loop, the zero TRIPLE of a three-dword store was three registers the lane-refill kernel kept in scratch)
                    asm volatile("" : "+v"(zero));
                    KC.normal[idx * 3 + 0] = zero;
                    KC.normal[idx * 3 + 1] = zero;
                    KC.normal[idx * 3 + 2] = zero;
                    KC.depth[idx] = 1e30f;
                    KC.object_id[idx] = -1;
                }
                if (KC.use_sky) { // sampleSky (render_utils.cuh:115-137): gradient, or the equirect map
                    if (KC.env) {
                        const float phi = det_atan2(rd.z, rd.x);
                        const float theta = det_acos(max_(-1.0f, min_(1.0f, rd.y)));
                        const float u = (phi + PI_F) * (1.0f / TWO_PI_F);
                        const float v = theta * (1.0f / PI_F);
                        acc = acc + throughput * tex2d_env(KC.env, KC.env_w, KC.env_h, u, v);
                    } else {
                        const float t = 0.5f * (rd.y + 1.0f);
                        acc = acc + throughput * lerp(KC.sky_bottom, KC.sky_top, t);
                    }
                } else {
                    acc = acc + throughput * mk3(0.0f);
                }
                end_path = true;
            } else {
                shaded = true;
                if (PMODE == 1) { // the triangle and the mesh's flags are in LDS
                    const int ti = (h.slot * 3 + h_order * PAIR_PAD) * 4;
                    const f3 gn = mk3(lds_ld1((const float *)PL.tris, ti + 3), lds_ld1((const float *)PL.tris, ti + 7),
                                      lds_ld1((const float *)PL.tris, ti + 11));
                    const int fl = __float_as_int(lds_ld1((const float *)PL.meshtab, h_order * 4 + 2));
                    hit = make_surface_of(KC, h, gn, fl, ro, rd, nullptr);
                } else {
                    hit = make_surface(KC, h, ro, rd, nullptr, nullptr);
                }
                if (sb == 0) {
                    const size_t idx = pidx(KC.width);
                    KC.normal[idx * 3 + 0] = hit.normal.x;
                    KC.normal[idx * 3 + 1] = hit.normal.y;
                    KC.normal[idx * 3 + 2] = hit.normal.z;
                    KC.depth[idx] = hit.t;
                    KC.object_id[idx] = h.mesh;
                }
                float4 m0, m2;
                if (mats_lds) {
                    m0 = lds_ld4(PL.mats, h_order * MAT_F4 + 0);
                    m2 = lds_ld4(PL.mats, h_order * MAT_F4 + 2);
                } else {
                    m0 = KC.materials[h.mesh * 6 + 0];
                    m2 = KC.materials[h.mesh * 6 + 2];
                }
                if (!hit.front_face) { // Beer-Lambert on back faces (path_logic.cuh:823-829)
                    const f3 T_unit = mk3(max_(1e-6f, m0.x), max_(1e-6f, m0.y), max_(1e-6f, m0.z));
                    const f3 absorption = mk3(-det_log(T_unit.x), -det_log(T_unit.y), -det_log(T_unit.z));
                    throughput = throughput * beerLambert(absorption, hit.t);
                }
                if (m2.x > 0.0f || m2.y > 0.0f || m2.z > 0.0f) {
                    if (sb < 0x10000 || prev_was_specular)
                        acc = acc + throughput * mk3(m2.x, m2.y, m2.z);
                }
                // light sample of next-event estimation (path_logic.cuh:305-382, 840)
                if (!ray_spec && KC.n_lights > 0) {
                    float r = rng_uniform(rng);
                    r = min_(r, 0.99999994f);
                    const int light_index = (int)(r * (float)KC.n_lights);
                    const LightRec light = lights_lds ? load_light<true>(PL.lights, light_index) : load_light(KC.lights, light_index);
                    const float pdf_pick = 1.0f / (float)KC.n_lights;
                    float attenuation = 1.0f;
                    float light_dist = 1e30f;
                    const f3 light_radiance = light.color * light.intensity;
                    if (light.type == 1) {
                        L = -light.direction;
                        pdf_sample = pdf_pick;
                    } else {
                        const f3 toLight = light.position - hit.point;
                        const float light_dist_sq = dot(toLight, toLight);
                        light_dist = sqrt_ieee(light_dist_sq);
                        if (light.radius <= 0.0f) {
                            L = toLight / light_dist;
                            pdf_sample = pdf_pick;
                        } else {
                            float sin_theta_max_sq = (light.radius * light.radius) / light_dist_sq;
                            sin_theta_max_sq = min_(sin_theta_max_sq, 0.9999f);
                            const float cos_theta_max = sqrt_ieee(1.0f - sin_theta_max_sq);
                            L = sample_cone_direction(rng, toLight / light_dist, cos_theta_max);
                            const float solid_angle = TWO_PI_F * (1.0f - cos_theta_max);
                            pdf_sample = (solid_angle > 1e-6f) ? (pdf_pick / solid_angle) : pdf_pick;
                        }
                        attenuation = attenuate(light_dist, light.range);
                        if (light.type == 2) {
                            const float theta = dot(L, -light.direction);
                            const float epsilon = light.inner - light.outer;
                            float spotIntensity;
                            if (epsilon <= 1e-6f)
                                spotIntensity = (theta >= light.outer) ? 1.0f : 0.0f;
                            else
                                spotIntensity = clampf((theta - light.outer) / epsilon, 0.0f, 1.0f);
                            attenuation *= spotIntensity;
                        }
                    }
                    const f3 shadow_offset = dot(hit.normal, L) > 0.0f ? hit.normal * 1e-4f : -hit.normal * 1e-4f;
                    shadow_o = hit.point + shadow_offset;
                    shadow_tmax = light_dist - 1e-3f;
                    // bsdf * light_radiance * attenuation / pdf: the last three factors are kept apart
                    // so the product is formed in the reference's order once visibility is known
                    light_scale = light_radiance;
                    light_att = attenuation;
                    want_shadow = true;
                }
            }
        }
        TS_LANES(19, shaded);
        TS_LANES(20, want_shadow);
        if (LDS_COUNT) {
            const unsigned long long add = (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act)) |
                                           ((unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(want_shadow)) << 32);
            if (lane == 0)
                __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned long long *)lds_count, add, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            n_shadow += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(want_shadow));
        }

        if (PMODE == 1 || TS_SHADING)
            TS_ADD(9, t_pc);
        PT_MARK("C2");
        phase_prio<PMODE, 3, 2>();
        const KParams &KC2 = kparams(kp0);
        const unsigned long long t_pc2 = TS_NOW();
        // ---- [C2] the light sample's value, BEFORE its visibility is known (path_logic.cuh:840-867: bsdf * radiance *
        // attenuation / pdf, soft clamp, MIS weight): `lit_now = throughput * direct * wgt` is what a visible sample adds
        // to `acc`, formed from the same operands in the same order as in the reference and added in the same place, so
        // no bit changes.  A sample that adds nothing either way (outside a spot cone, BSDF zero below the horizon)
        // needs no shadow ray: it is counted -- the reference traces it -- but not walked.
        bool lit = false;
        f3 lit_now = mk3(0.0f);
        if (want_shadow) {
            int mi = mats_lds ? h_order : h.mesh;
            asm volatile("" : "+v"(mi)); // (its own fetch of the material: not 22 registers live across the shadow phase)
            const Material mat = mats_lds ? load_material<true, MAT_F4>(PL.mats, mi) : load_material(KC2.materials, mi);
            const f3 V = -rd;
            const f3 bsdf = evaluateBSDF<FULL>(hit, mat, L, V);
            if (pdf_sample > 0.0f) {
                f3 direct = bsdf * light_scale * light_att / pdf_sample;
                direct = clamp_vector_soft(direct, 500.0f);
                if (direct.x > 0.0f || direct.y > 0.0f || direct.z > 0.0f) {
                    const float pdf_brdf = material_pdf<FULL>(hit, mat, V, L);
                    const float wgt = mis_weight(pdf_sample, pdf_brdf);
                    lit_now = throughput * direct * wgt;
                    lit = true;
                }
            }
        }

        // (SURVEY 8(d) counts rays actually traced: the samples of [C] that are not walked are counted apart -- rare, so
        // the LDS add sits behind a wave-uniform branch)
        {
            const unsigned long long zm = __builtin_amdgcn_ballot_w64(want_shadow && !lit);
            if (zm) {
                if (LDS_COUNT) {
                    if (lane == 0)
                        __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned long long *)(lds_count + 1),
                                               (unsigned long long)__builtin_popcountll(zm), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    n_zero += (uint32_t)__builtin_popcountll(zm);
                }
            }
        }
        if (PMODE == 1 || TS_SHADING)
            TS_ADD(10, t_pc2);
        TS_LANES(21, lit);
        PT_MARK("D");
        phase_prio<PMODE, 4, 3>();
        const KParams &KD = kparams(kp0);
        // ---- [D] shadow rays, all lanes that have one together (bvh_any_hit_tlas); PMODE 4 parks them instead and
        // walks them with the next extension rays
        const unsigned long long t_sh = TS_NOW();
        if (MERGED) {
            if (lit) {
                pend = lit_now;
                park_o = shadow_o;
                park_d = L;
                park_tmax = shadow_tmax;
                pending = true;
            }
        } else if (__builtin_amdgcn_ballot_w64(lit)) {
            const bool in_shadow = (PMODE == 1)   ? any_hit_pairs(KD, PL, lane, lit, shadow_o, L, shadow_tmax)
                                   : (PMODE == 2) ? any_hit_pairs_dyn(KD, PL, lane, lit, shadow_o, L, shadow_tmax)
                                   : (PMODE == 3) ? any_hit_pairs_tlas(KD, PL, lane, lit, shadow_o, L, shadow_tmax, cyc)
                                                  : any_hit<GEOM>(KD, lit, shadow_o, L, shadow_tmax, stk);
            if (lit && !in_shadow)
                acc = acc + lit_now;
        }
        TS_ADD(13, t_sh);

        PT_MARK("E");
        phase_prio<PMODE, 5, 4>();
        const KParams &KE = kparams(kp0);
        const unsigned long long t_pe = TS_NOW();
        // ---- [E] second half of the shading
        if (shaded) {
            const Material mat = mats_lds ? load_material<true, MAT_F4>(PL.mats, h_order) : load_material(KE.materials, h.mesh);
            f3 scatter_dir = mk3(0.0f), att = mk3(0.0f);
            bool is_specular = false;
            // (the path's last vertex -- the depth limit follows: the scattered ray is never traced and the throughput never read
            // again, so only the generator moves on, by the uniforms the reference draws here: lobe, direction, roulette.  With
            // the samples in step this is the whole wave in one iteration of max_depth.)
            const bool last = (sb >> 16) + 1 >= KE.max_depth;
            if (!material_scatter<FULL>(hit, mat, rd, rng, scatter_dir, att, is_specular, last)) {
                end_path = true;
            } else {
                prev_was_specular = is_specular;
                bool killed = false;
                if (sb >= 0x20000) { // Russian roulette (path_logic.cuh:871-880)
                    const float p = max_(0.05f, min_(0.95f, max_(throughput.x, max_(throughput.y, throughput.z))));
                    if (rng_uniform(rng) > p)
                        killed = true;
                    else if (!last)
                        throughput = throughput / p;
                }
                if (killed || last) {
                    end_path = true;
                } else {
                    throughput = throughput * att;
                    throughput = clamp_vector_soft(throughput, 50.0f);
                    const f3 off = hit.normal * 1e-4f;
                    ro = (dot(scatter_dir, hit.normal) > 0.0f) ? (hit.point + off) : (hit.point - off);
                    rd = scatter_dir;
                    ray_spec = is_specular;
                    sb += 0x10000; // ++bounce
                    if ((sb >> 16) >= KE.max_depth)
                        end_path = true;
                }
            }
        }
        if (act && end_path) {
            if (MERGED && pending) {
                fin = true; // closed after the next traversal, once the parked light sample is in
            } else {
                acc = clamp_vector_soft(acc, 100.0f);
                close_sample(acc);
                if (MERGED)
                    acc = mk3(0.0f);
            }
            ++sb; // ++s
            fresh = true;
        }
        if (PMODE == 1 || TS_SHADING)
            TS_ADD(11, t_pe);
    }

    PT_MARK("Z");
    const KParams &KZ = kparams(kp0);
    TS_ADD(14, t_kernel);
    cyc.flush(lane);
    if (inside) {
        const int x = px(), yl = pyl();
        const size_t idx = (size_t)yl * KZ.width + x, npix = KZ.rng_plane;
        KZ.rng[idx] = rng.d;
        KZ.rng[npix + idx] = rng.v0;
        KZ.rng[2 * npix + idx] = rng.v1;
        KZ.rng[3 * npix + idx] = rng.v2;
        KZ.rng[4 * npix + idx] = rng.v3;
        KZ.rng[5 * npix + idx] = rng.v4;
        const f3 out = avg_color / (float)KZ.spp;
        KZ.accum[idx * 3 + 0] = out.x;
        KZ.accum[idx * 3 + 1] = out.y;
        KZ.accum[idx * 3 + 2] = out.z;
        // (tonemap_kernel fused, below: RGB8, rows flipped within the tile (scene.cuh:2013-2015); skipped -- wave-uniform --
        // when a denoiser / bloom / up-scale stage follows and tonemaps its own result)
    }
    if (STREAM) { // the pixels left in [R]; the image is tonemapped by tonemap_tiles_kernel behind this launch
        if (KZ.counters) {
            if (LDS_COUNT) {
                wave_sync();
                const unsigned long long t = lds_count[0];
                n_ext = (uint32_t)t;
                n_shadow = (uint32_t)(t >> 32);
                n_zero = (uint32_t)lds_count[1];
            }
            if (lane == 0) { // (a slot per wave and launch: concurrent launches of a split frame do not share one)
                unsigned long long *w = KZ.counters + ((size_t)blockIdx.x * (KZ.split_n > 1 ? KZ.split_n : 1) + KZ.split_i) * COUNTER_WORDS;
                w[0] += (unsigned long long)n_ext;
                w[1] += (unsigned long long)n_shadow;
                w[2] += (unsigned long long)n_px * (unsigned long long)KZ.spp;
                w[3] += (unsigned long long)n_zero;
            }
        }
        // the last wave out leaves the queue as it found it (every other wave has drawn its last ticket)
        if (lane == 0 && atomicAdd(KZ.queue + 1, 1u) == gridDim.x - 1u) {
            atomicExch(KZ.queue, 0u);
            atomicExch(KZ.queue + 1, 0u);
        }
        return;
    }
    if (KZ.rgb8) {
        // A tile row is 8 pixels = 24 contiguous bytes of the bottom-up image.  Full tiles of a frame whose rows are
        // dword-aligned leave as six dwords per row: lane c < 6 of a row takes the (at most two) pixels its dword
        // spans out of their lanes' registers (ds_bpermute), instead of three byte stores per lane.
        const int x = px(), yl = pyl();
        unsigned char r8 = 0, g8 = 0, b8 = 0;
        if (inside)
            tonemap_pixel(avg_color / (float)KZ.spp, r8, g8, b8);
        const bool full_tile = (tx * 8 + 8 <= KZ.width) && (ty * 8 + 8 <= KZ.rows) && (KZ.width % 4 == 0) &&
                               (((size_t)KZ.rgb8 & 3u) == 0u);
        if (full_tile) {
            const uint32_t pix = (uint32_t)r8 | ((uint32_t)g8 << 8) | ((uint32_t)b8 << 16);
            const int c = lane & 7, first = (4 * c) / 3;                 // first pixel of dword c; byte offset in it: (4c) % 3
            const int src = (lane & ~7) | (first < 7 ? first : 7), src1 = (lane & ~7) | (first + 1 < 7 ? first + 1 : 7);
            const unsigned long long two = (unsigned long long)(uint32_t)__shfl((int)pix, src) |
                                           ((unsigned long long)(uint32_t)__shfl((int)pix, src1) << 24);
            if (c < 6) {
                uint32_t *row = (uint32_t *)(KZ.rgb8 + ((size_t)rgb8_row(KZ, yl) * KZ.width + (size_t)tx * 8) * 3);
                row[c] = (uint32_t)(two >> (8 * ((4 * c) % 3)));
            }
        } else if (inside) {
            const size_t o = ((size_t)rgb8_row(KZ, yl) * KZ.width + x) * 3;
            KZ.rgb8[o + 0] = r8;
            KZ.rgb8[o + 1] = g8;
            KZ.rgb8[o + 2] = b8;
        }
    }
    if (KZ.counters) {
        if (LDS_COUNT) {
            wave_sync();
            const unsigned long long t = lds_count[0];
            n_ext = (uint32_t)t;
            n_shadow = (uint32_t)(t >> 32);
            n_zero = (uint32_t)lds_count[1];
        }
        const uint32_t a = n_ext, b = n_shadow,
                       c = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(inside)) * (uint32_t)KZ.spp;
        // one slot of four counters per workgroup, plain read-modify-write (only this workgroup touches
        // it within a launch; launches are ordered).  Three atomics per wave on three shared addresses
        // serialised at the L2 atomic unit: 97 K of them took 1.2 ms per 1080p frame -- hidden behind a
        // 2.7-ms trace, but the whole cost of a light frame (1 spp, 1 bounce: 1.19 ms -> 0.17 ms).
        if (lane == 0) {
            unsigned long long *w = KZ.counters + (size_t)(ty * KZ.tiles_x + tx) * COUNTER_WORDS;
            w[0] += (unsigned long long)a;
            w[1] += (unsigned long long)b;
            w[2] += (unsigned long long)c;
            w[3] += (unsigned long long)n_zero;
        }
    }
}

// Lane refill (path_trace_kernel<.., STREAM = true>): the fused tonemap as a pass of its own behind the launch (same tiles, same mapping, same
// bytes): a persistent wave's pixels are not a tile's, so there is no tile epilogue to put it in.  One wave per 8x8 tile.
// PRIO (option "tm_prio" | 2): the pass shares the chip with the OTHER stream's persistent waves (frames overlap); its waves are a
// few hundred instructions each and everything behind them on their stream waits, so they take the issue slots first.
template <bool PRIO> __global__ __launch_bounds__(64) void tonemap_tiles_kernel(const KParams K) {
    if (PRIO)
        __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x, tile = blockIdx.x;
    const int tx = tile % K.tiles_x;
    const int ty = K.split_n > 1 ? (tile / K.tiles_x) * K.split_n + K.split_i : tile / K.tiles_x;
    const int x = tx * 8 + (lane & 7), yl = ty * 8 + (lane >> 3);
    const bool inside = x < K.width && yl < K.rows;
    unsigned char r8 = 0, g8 = 0, b8 = 0;
    if (inside) {
        const size_t idx = (size_t)yl * K.width + x;
        tonemap_pixel(mk3(K.accum[idx * 3 + 0], K.accum[idx * 3 + 1], K.accum[idx * 3 + 2]), r8, g8, b8);
    }
    const bool full_tile = (tx * 8 + 8 <= K.width) && (ty * 8 + 8 <= K.rows) && (K.width % 4 == 0) && (((size_t)K.rgb8 & 3u) == 0u);
    if (full_tile) { // six dwords per tile row (see path_trace_kernel's epilogue)
        const uint32_t pix = (uint32_t)r8 | ((uint32_t)g8 << 8) | ((uint32_t)b8 << 16);
        const int c = lane & 7, first = (4 * c) / 3;
        const int src = (lane & ~7) | (first < 7 ? first : 7), src1 = (lane & ~7) | (first + 1 < 7 ? first + 1 : 7);
        const unsigned long long two = (unsigned long long)(uint32_t)__shfl((int)pix, src) |
                                       ((unsigned long long)(uint32_t)__shfl((int)pix, src1) << 24);
        if (c < 6) {
            uint32_t *row = (uint32_t *)(K.rgb8 + ((size_t)rgb8_row(K, yl) * K.width + (size_t)tx * 8) * 3);
            row[c] = (uint32_t)(two >> (8 * ((4 * c) % 3)));
        }
    } else if (inside) {
        const size_t o = ((size_t)rgb8_row(K, yl) * K.width + x) * 3;
        K.rgb8[o + 0] = r8;
        K.rgb8[o + 1] = g8;
        K.rgb8[o + 2] = b8;
    }
}

} // namespace pt

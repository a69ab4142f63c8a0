// pt_merged.hip.h -- PMODE 4: ONE traversal phase per iteration of the render loop (included by pt_render.hip.h).
//
// The phase-synchronous loop of PMODE 2 runs two traversals per iteration: the closest hit of every live lane's
// extension ray, then the any-hit of the light samples' shadow rays.  On the showcase frame the second one works on
// ~10 rays per wave and still costs a pair build, three barriers and ~18 loop iterations with most lanes idle
// (profiles/r01e_lane_occupancy.txt: 166 K shadow phases against 274 K closest phases per frame).
//
// Here a light sample's shadow ray is traced TOGETHER with the path's next extension ray:
//   * [C]/[E] evaluate the light sample completely -- BSDF, MIS weight, `pend = throughput * direct * wgt` -- and park
//     it with its shadow ray; the next iteration's traversal answers the visibility, and `acc += pend` happens right
//     after that traversal, before anything else of the next bounce touches `acc`.  Every float is therefore
//     combined in the reference's order (path_logic.cuh:840-867) and the generator draws keep their order (the
//     shadow test draws nothing).  A light sample whose contribution is exactly zero (outside a spot cone, BSDF
//     zero below the horizon) changes nothing whether it is visible or not: its ray is counted (the reference
//     traces it) but not walked.
//   * a path that ends with a parked sample finishes one iteration later (`fin`), while the lane's next sample is
//     already on its way.
//   * the pair queue holds (ray, mesh) pairs of both kinds; a lane walks whatever it takes.  Closest-hit pairs keep
//     their per-pair limit and strict-`<` merge (E3); any-hit pairs are the same walk with a fixed limit that stops
//     at the first hit (order-free, E4) and may give the bottom of their stack to idle lanes.
#pragma once

namespace pt {

// pair entry: lane (6 bits) | kind (1 bit, 1 = shadow ray) | mesh order (9 bits)

// root-box tests of both rays of every lane against the meshes of the (single) TLAS leaf -> pair list.  Extension
// pairs always fit (64 per mesh); shadow pairs are appended mesh by mesh from `s_from` while the list has room and
// the first mesh that did not fit is returned in `s_next` (wave-uniform).
PT_DEV int build_pairs_merged(const KParams &K, const PairLds &L, int lane, bool first, bool ext, const RayO &we, bool sh,
                              const RayO &ws, float stmax, int s_from, int &s_next) {
    const int n_mesh = K.pair_meshes;
    uint16_t *pairs = (uint16_t *)L.pairs;
    int base = 0;
    float tE;
    if (first) {
        for (int i = 0; i < n_mesh; ++i) {
            const MeshHead mh = staged_mesh_head(L, i);
            bool hb;
            if (mh.flags & 1) {
                float ds;
                const RayO lr = local_ray(K, mh.mesh, we, ds);
                hb = ext && slab(mh.bmin, mh.bmax, lr, T_FAR, tE);
            } else {
                hb = ext && slab(mh.bmin, mh.bmax, we, T_FAR, tE);
            }
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(hb);
            if (hb)
                pairs[base + lane_prefix(bal)] = (uint16_t)((uint32_t)lane | ((uint32_t)i << 7));
            base += __builtin_popcountll(bal);
        }
    }
    int i = s_from;
    if (!__builtin_amdgcn_ballot_w64(sh)) // (no parked light sample in this wave, or all of them answered)
        i = n_mesh;
    for (; i < n_mesh; ++i) {
        if (base + 64 > K.pair_cap)
            break;
        const MeshHead mh = staged_mesh_head(L, i);
        if (mh.flags & 2) // transmission > 0.5: invisible to shadow rays (intersection.cuh:509-511)
            continue;
        bool hb;
        if (mh.flags & 1) {
            float ds;
            const RayO lr = local_ray(K, mh.mesh, ws, ds);
            hb = sh && slab(mh.bmin, mh.bmax, lr, stmax * ds, tE);
        } else {
            hb = sh && slab(mh.bmin, mh.bmax, ws, stmax, tE);
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(hb);
        if (hb)
            pairs[base + lane_prefix(bal)] = (uint16_t)((uint32_t)lane | 64u | ((uint32_t)i << 7));
        base += __builtin_popcountll(bal);
    }
    s_next = i;
    return base;
}

// Drains the mixed pair queue [0, P).  Afterwards L.best[r] = min over ray r's closest-hit pairs of
// {t bits, order << 24 | slot} and L.occ[r] != 0 iff one of its any-hit pairs found a hit.
// CSTEAL: closest-hit pairs may be stolen from as well -- verified subtree stealing, see run_closest_queue (pt_render.hip.h): a
// thief's hit in front of its own leaf box and equal distances from two walks of one pair mark the ray in L.dirty[0], and
// trace_merged traces marked rays again with CSTEAL off.
template <bool CSTEAL>
PT_DEV void run_merged_queue(const KParams &K, const PairLds &L, int lane, int P, f3 eo, f3 ed, f3 so, f3 sd, float stmax,
                             CycleAcc &cyc) {
    LdsStack stk{L.stack + lane};
    int next = 0;
    bool busy = false, active = false, isany = false, xf = false, thief = false;
    int cur = 0, sp = 0, bot = 0, r = 0, oi = 0, sb = -1, gen = 0, vic = 0;
    float dirScale = 1.0f, tb = T_FAR, tcur = 0.0f; // closest: the pair's running limit; any: the ray's fixed limit
    RayO pr = make_ray(mk3(0.0f), mk3(0.0f, 0.0f, 1.0f));
    TravStats ts;
    // next subtree of this lane's stack that can still matter (E1); an any-hit pair whose ray is already known
    // to be blocked drops what it has left
    auto pop = [&]() {
        active = false;
        if (isany && L.occ[r] != 0u)
            sp = bot;
        while (sp > bot) {
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
    const unsigned long long t_run = TS_NOW();
    for (;;) {
        TS_WAVE(7);
        const unsigned long long t_it = TS_NOW();
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!busy);
        const int n_idle = __builtin_popcountll(idle);
        if (next < P && (n_idle >= K.fetch_min || n_idle == 64)) {
            // the list is consumed from its END: the shadow pairs (appended last; the long walks, and the ones idle
            // lanes can steal from) start first
            const int p = next + lane_prefix(idle);
            const bool take = !busy && p < P;
            const uint32_t e = ((const uint16_t *)L.pairs)[take ? P - 1 - p : 0];
            const int src = (int)(e & 63u);
            const bool any = take && (e & 64u) != 0u;
            // the rays of the owner lane come out of its registers (every lane executes the shuffles: a source lane
            // must be active for ds_bpermute); the shadow ray only when some taker's pair names it
            f3 po = mk3(__shfl(eo.x, src), __shfl(eo.y, src), __shfl(eo.z, src));
            f3 pd = mk3(__shfl(ed.x, src), __shfl(ed.y, src), __shfl(ed.z, src));
            float ytm = 0.0f;
            if (__builtin_amdgcn_ballot_w64(any)) {
                const f3 yo = mk3(__shfl(so.x, src), __shfl(so.y, src), __shfl(so.z, src));
                const f3 yd = mk3(__shfl(sd.x, src), __shfl(sd.y, src), __shfl(sd.z, src));
                ytm = __shfl(stmax, src);
                po.x = any ? yo.x : po.x; // (component by component: a select of two structs becomes an indexed stack array)
                po.y = any ? yo.y : po.y;
                po.z = any ? yo.z : po.z;
                pd.x = any ? yd.x : pd.x;
                pd.y = any ? yd.y : pd.y;
                pd.z = any ? yd.z : pd.z;
            }
            if (take && !(any && L.occ[src] != 0u)) {
                r = src;
                oi = (int)(e >> 7);
                isany = any;
                const int4 mt = staged_mesh_entry(L, oi);
                pair_ray_from(K, mt, po, pd, dirScale);
                pr = make_ray(po, pd);
                xf = (mt.z & 1) != 0;
                tb = any ? (xf ? ytm * dirScale : ytm) : T_FAR;
                cur = mt.x;
                sp = bot = 0;
                sb = -1;
                thief = false;
                ++gen;
                busy = active = true;
            }
            next += n_idle;
        }
        if (!__builtin_amdgcn_ballot_w64(busy))
            break;
        // ---- subtree stealing among the any-hit pairs (see run_any_queue): once the queue is empty an idle lane
        // takes the BOTTOM stack entry of a busy any-hit lane together with a copy of its ray
        bool can_steal = false;
        const bool csteal = CSTEAL && K.csteal > 0; // (wave-uniform)
        if (csteal && next >= P && K.csteal_follow) { // a closest-hit thief follows its victim's limit while that walk lasts
            const int vl = vic & 63;
            const float vt = __shfl(tb, vl);
            const int vg = __shfl(gen, vl);
            if (busy && thief && !isany && vg == (vic >> 8) && vt < tb) {
                tb = vt;
                sb = -1;
            }
        }
        if ((K.steal || csteal) && next >= P) {
            const unsigned long long thieves = __builtin_amdgcn_ballot_w64(!busy);
            const bool is_victim = busy && (isany ? K.steal != 0 : csteal) && active && sp > bot;
            const unsigned long long victims = __builtin_amdgcn_ballot_w64(is_victim);
            if (thieves && victims) {
                const int nt = __builtin_popcountll(thieves), nv = __builtin_popcountll(victims);
                const int k = nt < nv ? nt : nv;
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
                const float ntb = __shfl(tb, v), nds = __shfl(dirScale, v);
                const int nr = __shfl(r, v), ng = __shfl(gen, v);
                const int noi = __shfl(oi | (xf ? 1 << 30 : 0) | (isany ? 1 << 29 : 0), v);
                if (steal) {
                    const uint2 e = L.stack[vb * 64 + v];
                    const float tE = __uint_as_float(e.y);
                    if (tE < ntb) { // (a closest-hit entry the reference culls as well; any-hit entries carry 0)
                        cur = (int)e.x;
                        tcur = tE;
                        npr.sx = npr.inv.x < 0;
                        npr.sy = npr.inv.y < 0;
                        npr.sz = npr.inv.z < 0;
                        pr = npr;
                        tb = ntb;
                        dirScale = nds;
                        r = nr;
                        oi = noi & ~(3 << 29);
                        xf = (noi >> 30) & 1;
                        isany = (noi >> 29) & 1;
                        sp = bot = 0;
                        sb = -1;
                        ++gen;
                        vic = v | (ng << 8);
                        thief = true;
                        busy = active = true;
                        if (!isany) {
                            TS_EVENT(0);
                        }
                    }
                }
                if (is_victim && vrank < k)
                    ++bot;
                wave_lds_order();
            }
            can_steal = thieves != 0ull && __builtin_amdgcn_ballot_w64(busy && active && (isany ? K.steal != 0 : csteal)) != 0ull;
        }
        const int yield_n = csteal ? K.csteal : K.steal, leaf_min = csteal ? K.csteal_leaf_min : K.leaf_min;
        // ---- inner nodes: wave-uniform loop with a predicated step; ends once K.leaf_min lanes wait at a leaf, or
        // after K.steal steps while idle lanes wait for stack entries to take
        TS_ADDQ(11, t_it);
        const unsigned long long t_nd = TS_NOW();
        int steps = 0;
        for (;;) {
            const bool innode = active && cur >= 0;
            if (!__builtin_amdgcn_ballot_w64(innode))
                break;
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(active && cur < 0)) >= leaf_min)
                break;
            if (can_steal && ++steps > yield_n)
                break;
            if (innode) {
                TS_WAVE(2);
                TS_LANE(3);
                if (PT_TWO_LEVEL && !CSTEAL) { // (any-hit pairs walk like closest-hit ones here, with a fixed limit)
                    if (descend2(K.nodes2, stk, sp, pr, tb, cur))
                        pop();
                    continue;
                }
                const float4 n0 = K.nodes[cur * 4 + 0], n1 = K.nodes[cur * 4 + 1], n2 = K.nodes[cur * 4 + 2],
                             n3 = K.nodes[cur * 4 + 3];
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
                    if (CSTEAL)
                        tcur = nearL ? tL : tR;
                } else {
                    pop();
                }
            }
        }
        TS_ADDQ(9, t_nd);
        const unsigned long long t_lf = TS_NOW();
        // ---- leaf phase as compacted (lane, triangle) tests (see run_closest_queue); a test sees its lane's limit
        // at leaf entry and merges with a 64-bit LDS min on {t bits, index in leaf}
        const bool atleaf = active && cur < 0;
        if (__builtin_amdgcn_ballot_w64(atleaf)) { // (an iteration that ended for the thieves' sake has no leaf to serve)
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
                if (key != ~0ull && isany) { // blocked: this ray needs nothing more
                    L.occ[r] = 1u;
                    active = false;
                } else {
                    if (key != ~0ull) {
                        tb = __uint_as_float((uint32_t)(key >> 32));
                        sb = first + (int)(uint32_t)key;
                        if (CSTEAL && thief && !(tcur < tb)) { // a hit in front of its own leaf box: the reference may never have come here
                            __hip_atomic_fetch_or(L.dirty, 1ull << r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            TS_EVENT(1);
                        }
                    }
                    pop();
                }
            }
        }
        TS_ADDQ(10, t_lf);
        if (busy && !active) { // this pair is finished: a closest-hit pair merges into its ray
            if (!isany && sb >= 0) {
                const float tw = xf ? tb / dirScale : tb;
                const unsigned long long key =
                    ((unsigned long long)__float_as_uint(tw) << 32) | ((unsigned long long)(uint32_t)oi << 24) | (uint32_t)sb;
                const unsigned long long old = __hip_atomic_fetch_min(&L.best[r], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (CSTEAL && (old >> 24) == (key >> 24) && old != key) { // two walks of one pair at the same distance
                    __hip_atomic_fetch_or(L.dirty, 1ull << r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    TS_EVENT(2);
                }
            }
            busy = false;
        }
    }
#ifdef PT_TRAV_STATS
    ts.v[0] = lane == 0 ? 1u : 0u;
    ts.v[1] = lane == 0 ? (unsigned)P : 0u;
#endif
    TS_ADDQ(8, t_run);
    ts.flush(0, lane, L.stat_bounce);
}

// extension ray (closest hit -> h) and parked shadow ray (any hit -> occluded) of every lane in one traversal
PT_DEV void trace_merged(const KParams &K, const PairLds &L, int lane, bool ext, f3 eo, f3 ed, bool sh, f3 so, f3 sd,
                         float stmax, Hit &h, bool &occluded, CycleAcc &cyc) {
    const RayO we = make_ray(eo, ed), ws = make_ray(so, sd);
    float tE;
    const bool ext_in = ext && slab(tlas_bmin(K), tlas_bmax(K), we, T_FAR, tE);
    const bool sh_in = sh && slab(tlas_bmin(K), tlas_bmax(K), ws, stmax, tE);
    L.best[lane] = ~0ull;
    L.occ[lane] = 0u;
    const int n_mesh = K.pair_meshes;
    int s_next = 0;
    bool first = true;
    const bool stealing = K.csteal > 0 && L.dirty; // (wave-uniform)
    if (stealing && lane == 0)
        L.dirty[0] = 0ull;
    for (;;) {
        const int s_from = s_next;
        const int P = build_pairs_merged(K, L, lane, first, ext_in, we, sh_in, ws, stmax, s_from, s_next);
        wave_sync();
        if (stealing)
            run_merged_queue<true>(K, L, lane, P, eo, ed, so, sd, stmax, cyc);
        else
            run_merged_queue<false>(K, L, lane, P, eo, ed, so, sd, stmax, cyc);
        wave_sync();
        first = false;
        // (the list is sized so that this is one pass unless nearly every ray touches nearly every mesh)
        if (s_next >= n_mesh || !__builtin_amdgcn_ballot_w64(sh_in && L.occ[lane] == 0u))
            break;
    }
    if (stealing) { // extension rays a thief could not vouch for: traced again, every pair walked by one lane
        const unsigned long long dm = L.dirty[0];
        wave_sync();
        if (dm) {
            const bool redo = ext_in && ((dm >> lane) & 1ull);
            if (redo) {
                TS_EVENT(3);
            }
            if (redo)
                L.best[lane] = ~0ull;
            int dummy = 0;
            const int P2 = build_pairs_merged(K, L, lane, true, redo, we, false, ws, stmax, n_mesh, dummy);
            wave_sync();
            run_merged_queue<false>(K, L, lane, P2, eo, ed, so, sd, stmax, cyc);
            wave_sync();
        }
    }
    const unsigned long long key = L.best[lane];
    occluded = sh && (L.occ[lane] != 0u);
    wave_sync(); // the lists are rebuilt by the next trace
    h.u = h.v = 0.0f;
    if (!ext || key == ~0ull) {
        h.t = h.t_local = T_FAR;
        h.mesh = -1;
        h.slot = -1;
        return;
    }
    const int4 mt = staged_mesh_entry(L, (int)((key >> 24) & 0xffu));
    h.t = __uint_as_float((uint32_t)(key >> 32));
    h.mesh = mt.w;
    h.slot = (int)(key & 0xffffffu);
    h.t_local = h.t;
    if (mt.z & 1)
        h.t_local = winner_t_local(K, h.mesh, h.slot, eo, ed);
}

} // namespace pt

// pt_build.hip.h -- GPU BVH (re)build of one mesh for dynamic geometry.
//
// Replaces the per-frame part of Scene::updateAccelerationStructures for a dirty mesh
// (scene.cuh:656-733): CPU Mesh::buildBVH (recursive median split, mesh.cuh:403-492) +
// Mesh::upload/uploadBVH reallocations (mesh.cuh:330-346, 494-516).
//
// Design.  The reference's builder always yields a balanced tree whose SHAPE (how many faces
// every node covers) depends only on the face count: a node of n > 17 faces splits into
// n/2 and n - n/2.  So for a mesh whose face count does not change, a rebuild never changes the
// shape -- only WHICH face sits in which leaf position.  The GPU build therefore keeps the
// uploaded topology and
//   1. centroid_bounds_kernel   min/max of the face centroids        (ordered-int atomics)
//   2. morton_kernel            30-bit Morton code of every centroid (10 bits per axis)
//   3. radix sort, 4 x 8 bits   (code, face) pairs, stable LSD: rs_hist / rs_scan / rs_scatter
//   4. apply_order_kernel       leaf position p of the mesh <- p-th face in Morton order
//   5. the refit pipeline of pt_refit.hip.h (packets, leaf boxes, level-by-level node boxes,
//      TLAS root box)
// i.e. an object-median split along the Z-order curve instead of along the longest centroid
// axis; depth, node count and leaf sizes are exactly the reference builder's, so the 24-entry
// traversal stack bound holds by construction (an LBVH's depth is data dependent).
// Everything stays on the context's stream; nothing returns to the host.
//
// Parity: closest-hit results do not depend on which valid BVH is traversed except for exact-t
// ties; the tests read the face order back (ptrt_read_prim_order), give the SAME
// tree to the oracle and compare frames bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pt {

constexpr int RS_WAVE_KEYS = 1024; // keys ranked by one wave (16 rounds of 64), the sort's unit of work
constexpr int RS_BLOCK = 256;      // 4 waves per workgroup

__device__ __forceinline__ uint32_t ordered_bits(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// cbounds[0..2] = min, [3..5] = max of the centroids, as ordered bits (memset to ff.. / 00.. before)
__global__ __launch_bounds__(256) void centroid_bounds_kernel(const float *__restrict__ verts,
                                                              const int4 *__restrict__ face_src, int n_faces,
                                                              float *__restrict__ centroids, uint32_t *cbounds) {
    // grid-stride: few workgroups, so the six global atomics per workgroup do not pile up
    uint32_t tmn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, tmx[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_faces; i += gridDim.x * blockDim.x) {
        const int4 f = face_src[i];
        for (int k = 0; k < 3; ++k) {
            // (a + b + c) * (1/3), the reference's centroid (mesh.cuh:425)
            const float c = (verts[f.x * 3 + k] + verts[f.y * 3 + k] + verts[f.z * 3 + k]) * (1.0f / 3.0f);
            centroids[(size_t)i * 3 + k] = c;
            const uint32_t o = ordered_bits(c);
            tmn[k] = o < tmn[k] ? o : tmn[k];
            tmx[k] = o > tmx[k] ? o : tmx[k];
        }
    }
    __shared__ uint32_t lo[3], hi[3];
    if (threadIdx.x < 3) {
        lo[threadIdx.x] = 0xffffffffu;
        hi[threadIdx.x] = 0u;
    }
    __syncthreads();
    for (int k = 0; k < 3; ++k) {
        uint32_t mn = tmn[k], mx = tmx[k];
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t a = __shfl_xor(mn, off), b = __shfl_xor(mx, off);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&lo[k], mn);
            atomicMax(&hi[k], mx);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        atomicMin(&cbounds[threadIdx.x], lo[threadIdx.x]);
        atomicMax(&cbounds[3 + threadIdx.x], hi[threadIdx.x]);
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) { // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void morton_kernel(const float *__restrict__ centroids, const uint32_t *__restrict__ cbounds,
                                                     int n_faces, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_faces)
        return;
    // one scale for the three axes (the largest extent): a flat mesh such as a water surface then
    // spends its code bits on the two axes it spans instead of on the noise of the third
    float ext = 0.0f;
    for (int k = 0; k < 3; ++k) {
        const float e = from_ordered_bits(cbounds[3 + k]) - from_ordered_bits(cbounds[k]);
        ext = e > ext ? e : ext;
    }
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = from_ordered_bits(cbounds[k]);
        float t = ext > 0.0f ? (centroids[(size_t)i * 3 + k] - lo) / ext : 0.0f;
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        int v = (int)(t * 1024.0f);
        q[k] = (uint32_t)(v > 1023 ? 1023 : v);
    }
    keys[i] = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
    vals[i] = (uint32_t)i;
}

// ---- stable LSD radix sort, 8 bits per pass.  A wave owns RS_WAVE_KEYS consecutive keys; the
// histogram is kept per wave so the scatter needs no inter-wave ordering inside a workgroup.
// hist layout: [wave][digit]; after rs_scan_kernel hist[w][d] is wave w's first output position
// for digit d (keys of smaller digits first, then the same digit in earlier waves).
__global__ __launch_bounds__(RS_BLOCK) void rs_hist_kernel(const uint32_t *__restrict__ keys, int n, int shift,
                                                           uint32_t *__restrict__ hist, int n_waves) {
    __shared__ uint32_t h[RS_BLOCK / 64][256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (RS_BLOCK / 64) + w;
    for (int d = lane; d < 256; d += 64)
        h[w][d] = 0;
    __syncthreads();
    if (gw < n_waves) {
        const int base = gw * RS_WAVE_KEYS;
        for (int r = 0; r < RS_WAVE_KEYS / 64; ++r) {
            const int i = base + r * 64 + lane;
            if (i < n)
                atomicAdd(&h[w][(keys[i] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (gw < n_waves)
        for (int d = lane; d < 256; d += 64)
            hist[(size_t)gw * 256 + d] = h[w][d];
}

// One workgroup: thread (seg, d) owns digit d over a quarter of the waves.  Column sums (loads
// independent of each other, coalesced over d), a 256-wide exclusive scan of the digit totals
// in LDS, then the columns are rewritten as running positions.
// (counts in `hist`, positions out to `pos`: distinct buffers, so the loads do not wait on the stores)
__global__ __launch_bounds__(1024) void rs_scan_kernel(const uint32_t *__restrict__ hist, uint32_t *__restrict__ pos,
                                                       int n_waves) {
    __shared__ uint32_t seg_total[4][256], digit_base[256], wave_sum[4];
    const int d = threadIdx.x & 255, seg = threadIdx.x >> 8;
    const int per = (n_waves + 3) / 4;
    const int w0 = seg * per, w1 = (w0 + per < n_waves) ? w0 + per : n_waves;
    uint32_t s = 0;
    for (int w = w0; w < w1; ++w)
        s += hist[(size_t)w * 256 + d];
    seg_total[seg][d] = s;
    __syncthreads();
    if (seg == 0) { // threads 0..255 = waves 0..3: exclusive scan of the 256 digit totals
        const uint32_t tot = seg_total[0][d] + seg_total[1][d] + seg_total[2][d] + seg_total[3][d];
        const int lane = d & 63;
        uint32_t inc = tot;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off);
            if (lane >= off)
                inc += t;
        }
        if (lane == 63)
            wave_sum[d >> 6] = inc;
        digit_base[d] = inc - tot;
    }
    __syncthreads();
    uint32_t run = digit_base[d];
    for (int k = 0; k < (d >> 6); ++k)
        run += wave_sum[k];
    for (int k = 0; k < seg; ++k)
        run += seg_total[k][d];
    for (int w = w0; w < w1; ++w) {
        const uint32_t v = hist[(size_t)w * 256 + d];
        pos[(size_t)w * 256 + d] = run;
        run += v;
    }
}

__global__ __launch_bounds__(RS_BLOCK) void rs_scatter_kernel(const uint32_t *__restrict__ keys_in,
                                                              const uint32_t *__restrict__ vals_in, int n, int shift,
                                                              const uint32_t *__restrict__ hist, int n_waves,
                                                              uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
    __shared__ uint32_t next[RS_BLOCK / 64][256]; // the wave's next output position per digit
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (RS_BLOCK / 64) + w;
    if (gw >= n_waves)
        return; // no workgroup-level synchronisation below
    for (int d = lane; d < 256; d += 64)
        next[w][d] = hist[(size_t)gw * 256 + d];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int base = gw * RS_WAVE_KEYS;
    for (int r = 0; r < RS_WAVE_KEYS / 64; ++r) {
        const int i = base + r * 64 + lane;
        const bool ok = i < n;
        const uint32_t key = ok ? keys_in[i] : 0u;
        const uint32_t val = ok ? vals_in[i] : 0u;
        const uint32_t digit = (key >> shift) & 255u;
        // lanes holding the same digit (valid lanes only): intersect the 8 per-bit ballots
        unsigned long long same = __builtin_amdgcn_ballot_w64(ok);
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64((digit >> b) & 1u);
            same &= ((digit >> b) & 1u) ? bal : ~bal;
        }
        const int rank = __builtin_popcountll(same & below);
        uint32_t pos = 0;
        if (ok)
            pos = next[w][digit] + (uint32_t)rank;
        __builtin_amdgcn_wave_barrier();
        if (ok && rank == 0)
            next[w][digit] += (uint32_t)__builtin_popcountll(same);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (ok) {
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
    }
}

// leaf slot s of the mesh holds prim position slot_pos[s]; give it the face ranked there
__global__ __launch_bounds__(256) void apply_order_kernel(const uint32_t *__restrict__ order, const int *__restrict__ slot_pos,
                                                          const int4 *__restrict__ face_src, int4 *__restrict__ slot_face,
                                                          int slot_base, int n_slots, uint32_t *cbounds) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6)
        cbounds[i] = i < 3 ? 0xffffffffu : 0u; // ready for the next build's atomics (saves two memset launches)
    if (i >= n_slots)
        return;
    const int s = slot_base + i;
    slot_face[s] = face_src[order[slot_pos[s]]];
}

// Triangle-soup meshes with a changing triangle count (updatePTScene's `Triangles` path,
// PTRTtransfer.cuh:2204-2385): the mesh keeps its uploaded capacity; triangles n_tris.. are
// degenerate copies of the last real vertex, which no ray can hit (|det| < 1e-6 rejects them,
// intersection.cuh:229) and which do not enlarge any box.
__global__ __launch_bounds__(256) void pad_soup_kernel(float *__restrict__ verts, int n_real_verts, int n_total_verts) {
    const int i = n_real_verts + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total_verts)
        return;
    const int src = n_real_verts > 0 ? n_real_verts - 1 : 0;
    for (int k = 0; k < 3; ++k)
        verts[(size_t)i * 3 + k] = n_real_verts > 0 ? verts[(size_t)src * 3 + k] : 0.0f;
}

} // namespace pt

// pt_refit.hip.h -- GPU BVH refit for dynamic geometry with unchanged topology (BASELINE
// config 5: "per-frame BVH refit + path trace").
//
// Replaces, for the per-frame part, the reference's Mesh::upload realloc + CPU Mesh::buildBVH +
// Mesh::uploadBVH + TLAS rebuild (mesh.cuh:330-346,403-516; scene.cuh:656-733) -- the reference
// has no refit.  Four small steps on the context's stream, no host synchronisation:
//   1. repack_tris_kernel    triangle packets {v0, e1, e2 | geometric normal} from the new vertices
//   2. refit_leaves_kernel   leaf boxes = min/max over the leaf's triangle vertices, stored into
//                            the parent's child-pair slot (or the mesh root box)
//   3. refit_level_kernel    one launch per WIDE tree level (> 2048 nodes), deepest first: node
//                            box = union of its two child boxes, stored into ITS parent's slot.
//                            A kernel boundary per level instead of in-kernel arrival counters:
//                            cross-XCD L2s are not coherent, and a level is microseconds of work.
//   4. refit_top_levels_kernel  the narrow levels near the root in ONE workgroup (barrier between
//                            levels; an empty launch costs ~4.6 us, a 13-level tree paid that 13x),
//                            then the mesh world boxes (Transform3D::transformAABB,
//                            transform.cuh:399-416) -> root box of the single-leaf TLAS
// All boxes are exact min/max of fp32 values, so a host refit of the same topology gives the
// same bits (Mesh::refitBVH in host/ptrt/mesh.hpp is what the oracle is fed).
#pragma once
#include "pt_kernels.hip.h"

namespace pt {

// Device-to-device copy of new vertex positions into the arena (ptrt_update_vertices): hipMemcpyAsync's blit took 93 us for the
// fluid scene's 4.7 MB (50 GB/s: profiles/r04_fluid_kernel_stats.csv), a tenth of the refit + trace frame; this is a plain
// grid-stride copy -- 16-byte lanes when both ends are 16-byte aligned, dwords otherwise.
__global__ __launch_bounds__(256) void copy_words_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t n_floats, int vec4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec4) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        const size_t n4 = n_floats / 4;
        for (size_t i = t; i < n4; i += stride)
            d4[i] = s4[i];
        for (size_t i = n4 * 4 + t; i < n_floats; i += stride)
            dst[i] = src[i];
    } else {
        for (size_t i = t; i < n_floats; i += stride)
            dst[i] = src[i];
    }
}

__global__ void repack_tris_kernel(const float *__restrict__ verts, const int4 *__restrict__ slot_face,
                                   float4 *__restrict__ tris, int n_slots) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots)
        return;
    const int4 f = slot_face[s];
    const float ax = verts[f.x * 3], ay = verts[f.x * 3 + 1], az = verts[f.x * 3 + 2];
    const float bx = verts[f.y * 3], by = verts[f.y * 3 + 1], bz = verts[f.y * 3 + 2];
    const float cx = verts[f.z * 3], cy = verts[f.z * 3 + 1], cz = verts[f.z * 3 + 2];
    const float4 p1 = make_float4(bx - ax, by - ay, bz - az, 0.0f), p2 = make_float4(cx - ax, cy - ay, cz - az, 0.0f);
    const f3 gn = packet_normal(p1, p2); // (the packets' w words: see tri_normals_kernel)
    tris[s * 3 + 0] = make_float4(ax, ay, az, gn.x);
    tris[s * 3 + 1] = make_float4(p1.x, p1.y, p1.z, gn.y);
    tris[s * 3 + 2] = make_float4(p2.x, p2.y, p2.z, gn.z);
}

// box -> its storage: dst >= 0: child slot (dst & 1) of inner node (dst >> 1); dst < 0: root box of mesh -dst-1
__device__ __forceinline__ void store_box(float4 *nodes, float4 *mesh_recs, int dst, float3 lo, float3 hi) {
    if (dst >= 0) {
        float *n = reinterpret_cast<float *>(nodes + (size_t)(dst >> 1) * 4) + ((dst & 1) ? 6 : 0);
        n[0] = lo.x; n[1] = lo.y; n[2] = lo.z;
        n[3] = hi.x; n[4] = hi.y; n[5] = hi.z;
    } else {
        float *r = reinterpret_cast<float *>(mesh_recs + (size_t)(-dst - 1) * MESH_REC_F4);
        r[0] = lo.x; r[1] = lo.y; r[2] = lo.z; // .w (root reference) untouched
        r[4] = hi.x; r[5] = hi.y; r[6] = hi.z; // .w (flags) untouched
    }
}

__global__ void refit_leaves_kernel(const float *__restrict__ verts, const int4 *__restrict__ slot_face,
                                    const int2 *__restrict__ leaves, const int *__restrict__ leaf_dst, float4 *nodes,
                                    float4 *mesh_recs, int n_leaves) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_leaves)
        return;
    const int2 lf = leaves[l];
    if (lf.y <= 0)
        return; // placeholder for an absent child: its box stays unhittable
    float3 lo = make_float3(1e30f, 1e30f, 1e30f), hi = make_float3(-1e30f, -1e30f, -1e30f);
    for (int i = 0; i < lf.y; ++i) {
        const int4 f = slot_face[lf.x + i];
        const int vi[3] = {f.x, f.y, f.z};
        for (int k = 0; k < 3; ++k) {
            const float x = verts[vi[k] * 3], y = verts[vi[k] * 3 + 1], z = verts[vi[k] * 3 + 2];
            lo.x = fminf(lo.x, x); lo.y = fminf(lo.y, y); lo.z = fminf(lo.z, z);
            hi.x = fmaxf(hi.x, x); hi.y = fmaxf(hi.y, y); hi.z = fmaxf(hi.z, z);
        }
    }
    store_box(nodes, mesh_recs, leaf_dst[l], lo, hi);
}

__global__ void refit_level_kernel(const int *__restrict__ level_nodes, int count, const int *__restrict__ node_dst,
                                   float4 *nodes, float4 *mesh_recs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    const int n = level_nodes[i];
    const float4 a = nodes[(size_t)n * 4 + 0], b = nodes[(size_t)n * 4 + 1], c = nodes[(size_t)n * 4 + 2];
    const float3 lo = make_float3(fminf(a.x, b.z), fminf(a.y, b.w), fminf(a.z, c.x));
    const float3 hi = make_float3(fmaxf(a.w, c.y), fmaxf(b.x, c.z), fmaxf(b.y, c.w));
    store_box(nodes, mesh_recs, node_dst[n], lo, hi);
}

// mesh world boxes (Transform3D::transformAABB, transform.cuh:399-416) -> root box of the single-leaf TLAS
__device__ inline void tlas_root_box(const float4 *mesh_recs, const int2 *tlas_leaves, const int *tlas_mesh_ids, int root_ref,
                                     float4 *root_box) {
    const int2 lf = tlas_leaves[~root_ref];
    float3 lo = make_float3(1e30f, 1e30f, 1e30f), hi = make_float3(-1e30f, -1e30f, -1e30f);
    for (int i = 0; i < lf.y; ++i) {
        const float4 *rec = mesh_recs + (size_t)tlas_mesh_ids[lf.x + i] * MESH_REC_F4;
        const float4 w0 = rec[5], w1 = rec[6], w2 = rec[7];
        for (int k = 0; k < 8; ++k) { // the 8 corners through the world matrix, as the host does
            const float x = (k & 1) ? rec[1].x : rec[0].x, y = (k & 2) ? rec[1].y : rec[0].y,
                        z = (k & 4) ? rec[1].z : rec[0].z;
            const float px = w0.x * x + w0.y * y + w0.z * z + w0.w;
            const float py = w1.x * x + w1.y * y + w1.z * z + w1.w;
            const float pz = w2.x * x + w2.y * y + w2.z * z + w2.w;
            lo.x = fminf(lo.x, px); lo.y = fminf(lo.y, py); lo.z = fminf(lo.z, pz);
            hi.x = fmaxf(hi.x, px); hi.y = fmaxf(hi.y, py); hi.z = fmaxf(hi.z, pz);
        }
    }
    root_box[0] = make_float4(lo.x, lo.y, lo.z, 0.0f);
    root_box[1] = make_float4(hi.x, hi.y, hi.z, 0.0f);
}

// The levels near the root hold few nodes each (2^(d-1)); one workgroup walks them all, deepest
// first, with a workgroup barrier between levels: a workgroup lives on one CU and shares its L1,
// so the barrier's workgroup-scope release/acquire makes a level's stores visible to the next.
// Thread 0 then derives the TLAS root box from the mesh root boxes just written.
struct TopLevels {
    int begin[24], count[24];
    int n;
};
__global__ __launch_bounds__(1024) void refit_top_levels_kernel(const int *__restrict__ level_nodes, TopLevels T,
                                                                const int *__restrict__ node_dst, float4 *nodes,
                                                                float4 *mesh_recs, const int2 *__restrict__ tlas_leaves,
                                                                const int *__restrict__ tlas_mesh_ids, int root_ref,
                                                                float4 *root_box) {
    for (int l = 0; l < T.n; ++l) {
        for (int i = threadIdx.x; i < T.count[l]; i += 1024) {
            const int n = level_nodes[T.begin[l] + i];
            const float4 a = nodes[(size_t)n * 4 + 0], b = nodes[(size_t)n * 4 + 1], c = nodes[(size_t)n * 4 + 2];
            const float3 lo = make_float3(fminf(a.x, b.z), fminf(a.y, b.w), fminf(a.z, c.x));
            const float3 hi = make_float3(fmaxf(a.w, c.y), fmaxf(b.x, c.z), fmaxf(b.y, c.w));
            store_box(nodes, mesh_recs, node_dst[n], lo, hi);
        }
        __syncthreads();
    }
    // (a TLAS with inner nodes is rebuilt by the host over the refitted mesh boxes: ptrt_update_instances)
    if (threadIdx.x == 0 && root_ref < 0)
        tlas_root_box(mesh_recs, tlas_leaves, tlas_mesh_ids, root_ref, root_box);
}

} // namespace pt

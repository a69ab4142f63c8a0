/*
 * ptrt.h -- C ABI of the MI355X (gfx950) path-tracing back end.
 *
 * This is the drop-in boundary for ONE hot path of Mark-Rindler/PTRT-game-engine:
 * the per-pixel path-tracing render loop behind `Scene::render_to_device`
 * (reference: src/pathtracer/scene/scene.cuh:1028-1209, kernel
 * src/pathtracer/scene/scene_kernels.cuh:122-194).  Every entry point below
 * names the reference interface it replaces.  Signatures are plain C: pointers,
 * sizes, ints.  No HIP, torch or C++ types cross this boundary.
 *
 * All functions returning `int` return PTRT_OK (0) on success or a negative
 * PTRT_E_* code; `ptrt_last_error(ctx)` gives the text.  The library never
 * falls back to a CPU implementation: with no usable HIP device every call that
 * needs one fails with PTRT_E_NO_DEVICE.
 *
 * Conventions shared with the reference:
 *   - HDR / G-buffers are row-major, y = 0 is the TOP of the view
 *     (scene_kernels.cuh:130-136); the RGB8 image is BOTTOM-UP, kernel row y is
 *     written to byte row H-1-y (scene.cuh:2013-2015).
 *   - material index == mesh index (path_logic.cuh:818-820).
 *   - one random stream per pixel, keyed by the GLOBAL pixel index y*W+x
 *     (scene_kernels.cuh:33-34), so a tile of a frame renders bit-identically
 *     to the same rows of the full frame.
 */
#ifndef PTRT_H
#define PTRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTRT_ABI_VERSION 6 /* 6: + ptrt_launch_ms_history (addition only); 2: ptrt_scene_desc gained env_rgba / env_width / env_height; 3: + ptrt_post_frame, ptrt_update_instances; 4: + ptrt_ring_*, ptrt_farm_* (additions only); 5: ptrt_stats gained shadow_rays_walked (the struct grew: rebuild callers of ptrt_get_stats) */

enum {
    PTRT_OK = 0,
    PTRT_E_INVALID = -1,   /* bad argument / inconsistent scene arrays        */
    PTRT_E_NO_DEVICE = -2, /* no HIP device or device index out of range      */
    PTRT_E_HIP = -3,       /* a HIP runtime call failed                       */
    PTRT_E_NOT_READY = -4, /* render before geometry/materials were uploaded  */
    PTRT_E_NOMEM = -5
};

/* ---- plain-data mirrors of the reference's device structs ---------------- */

typedef struct ptrt_vec3 { float x, y, z; } ptrt_vec3; /* common/vec3.cuh:8 (12 B) */

/* DeviceBVHNode, pathtracer/scene/mesh.cuh:37-43 (40 B).  Inner node: count==0,
 * left/right = child node indices.  Leaf: count>0, start = first slot in the
 * primitive-index array. */
typedef struct ptrt_bvh_node {
    ptrt_vec3 bmin, bmax;
    int32_t left, right, start, count;
} ptrt_bvh_node;

typedef struct ptrt_tri { int32_t v0, v1, v2; } ptrt_tri; /* mesh.cuh:45-47 */

/* Host-side description of one mesh: what DeviceMesh (math/intersection.cuh:90-106)
 * holds, with host pointers.  Matrices are the 16 floats of the reference's
 * mat4 exactly as Transform3D::updateMatrices leaves them
 * (scene/transform.cuh:260-306); the path reads them as ROW-major 3x4
 * (intersection.cuh:258-281). */
typedef struct ptrt_mesh_desc {
    const ptrt_vec3 *verts;
    int32_t vert_count;
    const ptrt_tri *faces;
    int32_t face_count;
    const ptrt_bvh_node *nodes; /* BLAS, pre-order, node 0 = root */
    int32_t node_count;
    const int32_t *prim_indices; /* leaf slots -> face index */
    int32_t prim_count;
    float world[16];
    float inverse[16];
    float normal[16];
    int32_t has_transform; /* scene.cuh:718-721 */
} ptrt_mesh_desc;

/* Light, pathtracer/scene/lights.cuh:14-26 (60 B, same field order). */
enum { PTRT_LIGHT_POINT = 0, PTRT_LIGHT_DIRECTIONAL = 1, PTRT_LIGHT_SPOT = 2 };
typedef struct ptrt_light {
    int32_t type;
    ptrt_vec3 position, direction, color;
    float intensity, range;
    float inner_cone, outer_cone; /* COSINES of the half angles (scene.cuh:1539-1540) */
    float radius;
} ptrt_light;

/* The ray-generation part of Camera (scene/camera.cuh:32-39). */
typedef struct ptrt_camera {
    ptrt_vec3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    float lens_radius;
} ptrt_camera;

/* DeviceMaterials, scene/material_lib.cuh:107-125: struct of 17 arrays, each
 * `count` long (host pointers here). */
typedef struct ptrt_materials {
    const ptrt_vec3 *albedo;
    const ptrt_vec3 *specular;
    const float *metallic;
    const float *roughness;
    const ptrt_vec3 *emission;
    const float *ior;
    const float *transmission;
    const float *transmission_roughness;
    const float *clearcoat;
    const float *clearcoat_roughness;
    const ptrt_vec3 *subsurface_color; /* carried, unused by the path */
    const float *subsurface_radius;    /* carried, unused by the path */
    const float *anisotropy;           /* carried, unused by the path */
    const float *sheen;
    const ptrt_vec3 *sheen_tint;
    const float *iridescence;
    const float *iridescence_thickness;
    int32_t count;
} ptrt_materials;

/* Everything `path_trace_kernel` receives that is not a frame buffer
 * (scene_kernels.cuh:123-129), flattened, host pointers. */
typedef struct ptrt_scene_desc {
    const ptrt_mesh_desc *meshes;
    int32_t mesh_count;
    const ptrt_bvh_node *tlas_nodes;
    int32_t tlas_node_count;
    const int32_t *tlas_mesh_indices;
    int32_t tlas_index_count;
    ptrt_materials materials;
    const ptrt_light *lights;
    int32_t light_count;
    ptrt_camera camera;
    ptrt_vec3 sky_top, sky_bottom;
    int32_t use_sky;
    /* Scene::loadHDRI's equirectangular map (scene.cuh:959-1026): env_width*env_height RGBA
     * floats, row 0 = what stbi_loadf returns first with flip_vertically_on_load(true);
     * NULL = none (`d_env_texture == 0`: the sky is the gradient).  ABI version 2. */
    const float *env_rgba;
    int32_t env_width, env_height;
} ptrt_scene_desc;

/* HitInfo, math/intersection.cuh:108-124, as returned by Scene::traceSingleRay. */
typedef struct ptrt_hit {
    int32_t hit;
    float t;
    ptrt_vec3 point, normal;
    int32_t mesh_index;
    int32_t front_face;
    float u, v;
    int32_t face_index;
    ptrt_vec3 local_point;
} ptrt_hit;

/* Per-frame counters of the trace stage (Mrays/s numerator). */
typedef struct ptrt_stats {
    uint64_t extension_rays; /* traceRay calls   (intersection.cuh:526) */
    uint64_t shadow_rays;    /* bvh_any_hit_tlas (intersection.cuh:481) */
    uint64_t paths;          /* pixel-samples started                   */
    /* ABI 5.  shadow_rays counts every light sample the reference path sends a shadow ray for (path_logic.cuh:357);
     * a sample whose value is exactly zero whatever its visibility (outside a spot cone, BSDF zero below the horizon)
     * is counted there but its ray is NOT walked by path_trace_kernel.  shadow_rays_walked excludes those, so
     * extension_rays + shadow_rays_walked = rays actually traced (SURVEY 8(d)'s Mrays/s numerator). */
    uint64_t shadow_rays_walked;
} ptrt_stats;

#define PTRT_BLUE_NOISE_SIZE 64 /* common/bluenoise.cuh: 64 x 64 x 2 floats */
#define PTRT_BLUE_NOISE_FLOATS (PTRT_BLUE_NOISE_SIZE * PTRT_BLUE_NOISE_SIZE * 2)
#define PTRT_DEFAULT_SEED 12345ULL /* scene.cuh:448 */

/* buffer kinds for ptrt_read_buffer / ptrt_device_buffer (tile rows only) */
enum {
    PTRT_BUF_ACCUM = 0,     /* float[rows*W*3]  HDR radiance, Scene::getNoisyColorBuffer (scene.cuh:1722) */
    PTRT_BUF_NORMAL = 1,    /* float[rows*W*3]  first-hit normal, getNormalBuffer (scene.cuh:1723)         */
    PTRT_BUF_DEPTH = 2,     /* float[rows*W]    first-hit t,      getDepthBuffer  (scene.cuh:1724)         */
    PTRT_BUF_OBJECT_ID = 3, /* int32[rows*W]    first-hit mesh index (scene_kernels.cuh:193)               */
    PTRT_BUF_RGB8 = 4,      /* uint8[rows*W*3]  tonemapped tile, bottom-up WITHIN the tile                 */
    PTRT_BUF_RNG = 5,       /* uint32[rows*W*6] generator state {d, v0..v4} per pixel (canonical order)     */
    PTRT_BUF_DENOISED = 6,  /* float[H*W*3]     denoiser output (d_denoised_buffer, scene.cuh:1121); denoiser on */
    PTRT_BUF_MOTION = 7,    /* float[H*W*2]     uv motion vectors, Scene::getMotionVectorBuffer (scene.cuh:1725)  */
    PTRT_BUF_RENDER_ACCUM = 8 /* float[rh*rw*3] the path tracer's colour image at the render size: d_scaled_accum
                                 (scene.cuh:181) after ptrt_set_render_size, else the same memory as ACCUM.  With a
                                 reduced render size NORMAL/DEPTH/OBJECT_ID/DENOISED/MOTION are rw x rh as well
                                 (the d_scaled_* set) and ACCUM holds the up-scaled final HDR image (scene.cuh:1194). */
};

typedef struct ptrt_ctx ptrt_ctx;

/* Scene::Scene(w,h) (scene.cuh:747-832): allocates frame buffers and the per-pixel
 * generator states on HIP device `device`.  The context renders rows
 * [tile_y0, tile_y0+tile_rows) of a full_w x full_h frame (tile_rows <= 0 means
 * the whole frame).  Installs nothing else: blue noise, RNG seed, scene are
 * separate calls. */
int ptrt_create(int full_w, int full_h, int tile_y0, int tile_rows, int device, ptrt_ctx **out);

/* Scene::~Scene (scene.cuh:834-941).  NULL and repeated destroy of a dead handle's
 * slot are tolerated the way the reference tolerates double cudaFree. */
void ptrt_destroy(ptrt_ctx *ctx);

/* The same for a context that owns every `period`-th 8-row strip of the frame, starting with strip `phase`
 * (strip t = rows 8t .. 8t+7): the tile farm's other way of cutting a frame (SURVEY 8(e) "interleaved strips ... to
 * balance sky vs geometry").  Contiguous bands of the showcase frame carry 1.6x the mean number of rays in the worst
 * band (three of eight bands are sky); interleaved, every context samples the whole frame.  The context's buffers
 * hold its strips one after the other (top-down; the RGB8 image bottom-up within them, like a band's);
 * period == 1 is the whole frame.  ABI 4. */
int ptrt_create_interleaved(int full_w, int full_h, int phase, int period, int device, ptrt_ctx **out);

const char *ptrt_last_error(const ptrt_ctx *ctx); /* never NULL */
int ptrt_abi_version(void);

/* initBlueNoise() -> cudaMemcpyToSymbol(d_blue_noise) (common/bluenoise.cuh:189-198).
 * `table` = 64*64*2 floats, [y][x][channel]. */
int ptrt_set_blue_noise(ptrt_ctx *ctx, const float *table);

/* Scene::initRandomStates + init_curand_kernel (scene.cuh:433-456,
 * scene_kernels.cuh:26-35): XORWOW state for every pixel of the tile,
 * seed `seed`, subsequence = global pixel index, offset 0. */
int ptrt_reset_rng(ptrt_ctx *ctx, unsigned long long seed);

/* Mesh::upload + Mesh::uploadBVH + Scene::buildAndUploadTLAS's copies + the
 * DeviceMesh descriptor upload (mesh.cuh:330-346,499-516; scene.cuh:458-594,
 * 684-727).  Host arrays are copied (and re-laid-out for the GPU) before return. */
int ptrt_upload_geometry(ptrt_ctx *ctx, const ptrt_mesh_desc *meshes, int mesh_count,
                         const ptrt_bvh_node *tlas_nodes, int tlas_node_count,
                         const int32_t *tlas_mesh_indices, int tlas_index_count);

/* Instances moved, nothing else changed -- the transform-dirty case of Scene::updateAccelerationStructures
 * (scene.cuh:656-743) and of commitObjectChanges (scene.cuh:1784-1787): takes the new world / inverse /
 * normal matrices and has_transform flags of every mesh and the rebuilt TLAS; vertices, BLASes and triangle
 * packets on the device stay where they are (a full ptrt_upload_geometry re-lays-out every triangle).
 * mesh_count must equal the uploaded one.  Read from each descriptor: world, inverse, normal, has_transform -- nothing
 * else (vertex, face and BVH pointers may be stale or NULL).  The meshes' root boxes on the device, which a ptrt_refit /
 * ptrt_build_bvh may have moved since the upload, are left alone; the world-space first-pass boxes of the instances
 * (scenes with a real TLAS) are recomputed from those DEVICE boxes (one small read-back; the call synchronises). */
int ptrt_update_instances(ptrt_ctx *ctx, const ptrt_mesh_desc *meshes, int mesh_count,
                          const ptrt_bvh_node *tlas_nodes, int tlas_node_count,
                          const int32_t *tlas_mesh_indices, int tlas_index_count);

/* Scene::uploadMaterialSoA (scene.cuh:286-431). */
int ptrt_upload_materials(ptrt_ctx *ctx, const ptrt_materials *mats);

/* the d_lights upload in Scene::updateAccelerationStructures (scene.cuh:636-651). */
int ptrt_upload_lights(ptrt_ctx *ctx, const ptrt_light *lights, int light_count);

/* `Camera cam` kernel argument (scene_kernels.cuh:124), set by Scene::setCamera (scene.cuh:1298). */
int ptrt_set_camera(ptrt_ctx *ctx, const ptrt_camera *cam);

/* Scene::setSkyGradient / disableSky (scene.cuh:1548-1565). */
int ptrt_set_sky(ptrt_ctx *ctx, const ptrt_vec3 *top, const ptrt_vec3 *bottom, int use_sky);

/* Scene::loadHDRI's device side (scene.cuh:976-1022): the equirectangular environment map that
 * sampleSky reads with tex2D<float4>(envMap, u, v) (render_utils.cuh:115-137) -- normalised
 * coordinates, address mode wrap in u / clamp in v, linear filtering.  `rgba`: width*height*4
 * floats (host pointer, copied); NULL frees the map (freeHDRI) and the sky is the gradient
 * again.  MI355X has no texture-filtering path worth a detour for one fetch per escaped ray: the
 * bilinear fetch is restated in the kernel with the CUDA texture unit's documented arithmetic
 * (xB = u*W - 0.5, weights quantised to 8 fractional bits).  SURVEY 8(f) rank 4. */
int ptrt_set_env_map(ptrt_ctx *ctx, const float *rgba, int width, int height);

/* Dynamic geometry, same topology (the `Triangles` path of updatePTScene,
 * src/common/PTRTtransfer.cuh:2249-2270, followed by Scene::commitObjectChanges,
 * scene.cuh:1784).  The reference re-allocates the mesh's device arrays and REBUILDS its BVH on
 * the CPU every frame (mesh.cuh:330-346,403-516); here the new vertex positions of mesh
 * `mesh_index` are copied into the arena (`verts`: vert_count x 3 floats, a DEVICE pointer if
 * verts_on_device != 0) and ptrt_refit() re-derives, on the GPU and on the context's stream,
 * the triangle packets, every leaf and inner box of the dirty meshes (bottom-up over the
 * UNCHANGED tree), the mesh root boxes and the TLAS root box.  No host synchronisation.
 * With a TLAS that has inner nodes (more meshes than one TLAS leaf holds) the TLAS itself is left to the
 * caller: rebuild it over the moved meshes' boxes, as the reference's commit does, and hand it to
 * ptrt_update_instances (the Scene mirror's refitObjectChanges / rebuildObjectChanges do).  A refitted tree has the boxes a
 * host refit of the same topology gives (min/max are exact), so results stay bit-comparable
 * with the oracle run on those arrays; it is NOT the tree a fresh median-split build would give.
 * Host positions (verts_on_device == 0): the caller's buffer is its own again when the call returns, and the call does NOT
 * wait for the stream's earlier work (the previous frame), so the host prepares frame N + 1 while the GPU renders frame N.
 * Ordinary memory is copied into pinned staging memory of the context and crosses PCIe asynchronously on the stream; memory
 * HIP knows as pinned crosses on a copy stream of the context into device staging (the call waits for that transfer alone)
 * and moves into the arena on the stream. */
int ptrt_update_vertices(ptrt_ctx *ctx, int mesh_index, const float *verts, int vert_count, int verts_on_device);
int ptrt_refit(ptrt_ctx *ctx);

/* GPU BVH (re)build of one mesh -- what Scene::updateAccelerationStructures does on the CPU for
 * every dirty mesh (scene.cuh:656-733: Mesh::buildBVH mesh.cuh:403-492 + upload/uploadBVH
 * mesh.cuh:330-346,494-516), for a mesh whose FACE COUNT is unchanged.
 * The reference builder's tree shape depends only on the face count (n > leafTarget+tol splits
 * into n/2 and n-n/2), so the uploaded topology is kept and the faces are re-assigned to the
 * leaf positions in Morton order of their current centroids (30-bit codes, stable radix sort on
 * the GPU), followed by ptrt_refit().  All on the context's stream, nothing returns to the host.
 * Depth and leaf sizes are the uploaded tree's, so the 24-entry stack bound keeps holding.
 * Needs: every face of the mesh in exactly one leaf position (any tree Mesh::buildBVH makes),
 * single-leaf TLAS.  ptrt_read_prim_order returns the resulting `primIndices` (mesh.cuh:57) so a
 * host copy of the tree (and the oracle) can follow. */
int ptrt_build_bvh(ptrt_ctx *ctx, int mesh_index);
int ptrt_read_prim_order(ptrt_ctx *ctx, int mesh_index, int32_t *prim_indices_out, int count);

/* The `Triangles` path of updatePTScene with a CHANGING triangle count (PTRTtransfer.cuh:2204-2385:
 * a new triangle list every frame, e.g. a fluid surface).  For a triangle-soup mesh (face i =
 * vertices 3i,3i+1,3i+2, what Scene::addTriangles makes) uploaded with room for N triangles:
 * copies tri_count <= N triangles (9 floats each; a DEVICE pointer if on_device != 0) and turns the
 * remaining N - tri_count into degenerate copies of the last real vertex -- never hit
 * (|det| < EPSILON, intersection.cuh:229), never enlarging a box.  Follow with ptrt_build_bvh(). */
int ptrt_update_triangles(ptrt_ctx *ctx, int mesh_index, const float *verts9, int tri_count, int on_device);

/* ---- post-process "next" row: motion vectors + spatiotemporal denoiser ----
 * DenoiserSettings of the non-split path (src/pathtracer/rendering/denoiser.cuh:36-73); the
 * defaults are the reference's diffuse_* values. */
typedef struct ptrt_denoiser_settings {
    float tau, min_alpha, max_history, sigma_luminance, sigma_normal, sigma_depth;
    int32_t atrous_iterations;
    float clamp_scale, firefly_threshold;
    float depth_reject_absolute, depth_reject_relative, normal_reject_threshold, sky_depth_threshold;
    float edge_depth_threshold, edge_normal_threshold;
    int32_t use_object_ids, enable_firefly_suppression;
} ptrt_denoiser_settings;
void ptrt_denoiser_default_settings(ptrt_denoiser_settings *out);

/* `new Denoiser(settings)` (denoiser.cuh:808-845; created in Scene::updateScaledBuffers,
 * scene.cuh:1978-1993): allocates history + scratch images, marks the next frame as the first.
 * While enabled, ptrt_render runs motion_vector_kernel (denoiser_kernels.cuh:33) with the matrix
 * given to ptrt_set_prev_view_proj, then Denoiser::denoise (denoiser.cuh:966-1064, non-split
 * path), and tonemaps the DENOISED image (scene.cuh:1103-1127,1204).  Full-frame contexts only
 * (the filters read up to 32 pixels across band borders).  settings == NULL: defaults. */
int ptrt_denoiser_enable(ptrt_ctx *ctx, const ptrt_denoiser_settings *settings);
/* Denoiser::destroy + delete (scene.cuh:1970-1976) */
int ptrt_denoiser_disable(ptrt_ctx *ctx);
/* Scene::prev_view_proj (scene.cuh:113,1208,1282): proj*view of the PREVIOUS frame, 16 floats as
 * mat4 stores them (column-major), consumed by the next ptrt_render's motion-vector pass. */
int ptrt_set_prev_view_proj(ptrt_ctx *ctx, const float *m16);

/* ---- post-process "next" row: bloom + resolution scale ----
 * perfSettings.enableBloom (scene.cuh:1137-1183): bright pass (threshold 1.5, knee 0.5), six
 * levels of 5-tap horizontal blur + 5-tap vertical down-sample, five bilinear up-sample-adds,
 * and the last up-sample-add INTO the frame's HDR image (the noisy colour buffer, or the denoised
 * one when the denoiser runs), before the tonemap.  Same arithmetic and the same mip sizes as
 * the reference's host loop, in 8 launches instead of 19 (pt_post.hip.h).  Needs a full-frame
 * context of at least 64x64 (below that a mip level is empty and the reference reads NULL). */
int ptrt_set_bloom(ptrt_ctx *ctx, int enabled);
/* Scene::updateScaledBuffers (scene.cuh:1913-2000) for perfSettings.resolutionScale: path trace,
 * motion vectors, denoiser and bloom run at render_w x render_h (the d_scaled_* buffers; pixel p
 * of the small frame continues generator state p, as the reference's launch does), then
 * upscale_bilinear_kernel (scene_kernels.cuh:406-441) fills the full-size colour buffer, which is
 * tonemapped.  Changing the size frees the denoiser (re-enable it; the reference re-creates it at
 * the new size) -- the caller also restarts accumulation.  The Scene wrapper computes
 * render_w = max(64, int(W * scale)) like the reference.  Full-frame contexts only. */
int ptrt_set_render_size(ptrt_ctx *ctx, int render_w, int render_h);

/* ---- presentation "next" row: the HIP half of rtgl::* ----
 * The reference presents through a CUDA-mapped GL pixel-buffer object
 * (src/common/glfw_view_interop.hpp:281-332: map_pbo_device_ptr -> render_to_device -> unmap_pbo
 * -> blit_pbo_to_texture -> draw_interop).  MI355X has no GL interop, so the PBO becomes a ring of
 * `slots` device RGB8 frames, each mirrored into PINNED host memory:
 *   ptrt_present_map(slot)      = map_pbo_device_ptr: the device pointer to render into (first waits
 *                                 until the slot's previous download has finished)
 *   ptrt_present_unmap(slot)    = unmap_pbo: enqueues the asynchronous device->host copy of the frame
 *                                 on the context's stream and records the slot's event; does not block
 *   ptrt_present_acquire(slot)  = what blit_pbo_to_texture needs: waits for that event and returns the
 *                                 pinned host pixels (W*H*3, bottom-up) for glTexSubImage2D / a file
 * With slots >= 2 the copy of frame i overlaps the rendering of frame i+1 (the viewer maps slot
 * (i+1)%slots while frame i is in flight).  Full-frame or band contexts alike. */
int ptrt_present_create(ptrt_ctx *ctx, int slots);
int ptrt_present_map(ptrt_ctx *ctx, int slot, void **device_pixels);
int ptrt_present_unmap(ptrt_ctx *ctx, int slot);
int ptrt_present_acquire(ptrt_ctx *ctx, int slot, const unsigned char **host_pixels);
int ptrt_present_destroy(ptrt_ctx *ctx);

/* The same ring WITHOUT a context: what rtgl::init_interop_viewer(V, width, height, title, cudaDevice)
 * (glfw_view_interop.hpp:174-279) creates before any Scene exists -- the GL pixel-buffer object registered with
 * CUDA becomes `slots` device frames of `frame_bytes` on `device`, each mirrored into pinned host memory.
 *   ptrt_ring_map     = cudaGraphicsMapResources + GetMappedPointer (glfw_view_interop.hpp:281-298)
 *   ptrt_ring_unmap   = cudaGraphicsUnmapResources (:300-307): enqueues the device->host copy behind the frame.
 *                       ptrt_render / ptrt_post_frame recognise a ring slot as their out_rgb8 and record the
 *                       slot's event on THEIR stream, so the copy waits for exactly that frame and the call site
 *                       stays `map -> scene.render_to_device(ptr) -> unmap`; a slot written by anything else is
 *                       ordered behind all work submitted to the device's blocking streams (NULL-stream event).
 *   ptrt_ring_acquire = what blit_pbo_to_texture (:309-317) needs: waits for the copy, returns the host pixels.
 * Errors: ptrt_last_error(NULL). */
typedef struct ptrt_ring ptrt_ring;
int ptrt_ring_create(int device, size_t frame_bytes, int slots, ptrt_ring **out);
int ptrt_ring_map(ptrt_ring *ring, int slot, void **device_pixels);
int ptrt_ring_unmap(ptrt_ring *ring, int slot);
int ptrt_ring_acquire(ptrt_ring *ring, int slot, const unsigned char **host_pixels);
void ptrt_ring_destroy(ptrt_ring *ring);

/* convenience: the five uploads above from one flattened description */
int ptrt_upload_scene(ptrt_ctx *ctx, const ptrt_scene_desc *scene);

/* The body of Scene::render_to_device (scene.cuh:1028-1209): path_trace_kernel, then -- for
 * full-frame contexts that enabled them -- motion vectors + denoiser, bloom, up-scale, and the
 * tonemap of the final HDR image (fused into whichever stage produces it).  `frame_index` is the reference's frame_count_ (jitter index frame+s,
 * scene_kernels.cuh:152-157; >= 0 -- a negative one is PTRT_E_INVALID: it would index the jitter table out of bounds).  `out_rgb8`: tile_rows*W*3 bytes, bottom-up within
 * the tile; a DEVICE pointer if out_is_device != 0 (the mapped PBO of
 * glfw_view_interop.hpp:281), else a host buffer (synchronous copy).  NULL skips
 * the copy (the RGB8 image stays readable through PTRT_BUF_RGB8).
 * Asynchronous w.r.t. the host when out_is_device != 0 or out_rgb8 == NULL, like
 * the reference (it returns right after the tonemap launch).  Ordering on the context's stream: see option "pipeline" of
 * ptrt_set_option -- the frame is complete before anything that follows the call on the stream, always.
 * out_is_device == PTRT_OUT_DEVICE_FRAME (ABI 5): out_rgb8 is the WHOLE width*height*3 frame (bottom-up) on the context's
 * device and a band / strip context writes its rows where they belong in it -- several contexts on one device fill one
 * frame without a copy (the tile farm does this for the parts on the presenting device).  Not with the denoiser, bloom or a
 * reduced render size; PTRT_BUF_RGB8 then has nothing to read; a presentation-ring slot used this way is marked by the caller
 * that joins the contexts (ptrt_farm_*), not by ptrt_render. */
#define PTRT_OUT_HOST 0
#define PTRT_OUT_DEVICE 1
#define PTRT_OUT_DEVICE_FRAME 2
int ptrt_render(ptrt_ctx *ctx, int frame_index, int spp, int max_depth, void *out_rgb8,
                int out_is_device);

/* cudaDeviceSynchronize at the call sites that have one (scene.cuh:455,1244). */
int ptrt_sync(ptrt_ctx *ctx);

/* synchronising read-back of one tile buffer into host memory */
int ptrt_read_buffer(ptrt_ctx *ctx, int kind, void *host_dst, size_t dst_bytes);

/* Scene::getNoisyColorBuffer/getNormalBuffer/getDepthBuffer (scene.cuh:1722-1725):
 * raw device pointer of a tile buffer (NULL on error).  PTRT_BUF_RNG is not
 * exposed this way (its device layout is private). */
void *ptrt_device_buffer(ptrt_ctx *ctx, int kind);

/* write generator states back in canonical order (tests / checkpointing) */
int ptrt_write_rng(ptrt_ctx *ctx, const uint32_t *states, size_t bytes);

/* Scene::traceSingleRay -> trace_single_ray_kernel (scene.cuh:1367-1391,
 * scene_kernels.cuh:38-49); batched: n rays, origins/directions as n*3 floats. */
int ptrt_trace_rays(ptrt_ctx *ctx, const float *origins, const float *directions, int n,
                    ptrt_hit *out_hits);

/* counters accumulated by ptrt_render since the last call (reset on read);
 * only maintained when ptrt_set_option(ctx,"count_rays",1). */
int ptrt_get_stats(ptrt_ctx *ctx, ptrt_stats *out);

/* Tile farm, presenting rank: steps 3-7 of Scene::render_to_device (motion vectors,
 * Denoiser::denoise, bloom, tonemap; scene.cuh:1103-1208) of a FULL-FRAME context over a frame whose
 * HDR image and G-buffers were rendered by other contexts (bands) and gathered into DEVICE memory:
 * accum/normal width*height*3 floats, depth width*height floats, object_id width*height ints, top-down,
 * i.e. the bands' ptrt_device_buffer contents concatenated in row order.  Camera, previous view-projection
 * and the denoiser/bloom switches are the context's own.  PTRT_E_INVALID for a band context, a reduced
 * render size, or when neither stage is enabled. */
int ptrt_post_frame(ptrt_ctx *ctx, const float *accum, const float *normal, const float *depth,
                    const int32_t *object_id, void *out_rgb8, int out_is_device);

/* ---- the tile farm below the C ABI: one process, the GPUs of one node --------------------------------
 * The reference has no multi-GPU code; a C++ application built on the Scene mirror farms a frame by holding one
 * band / strip context per GPU (ptrt_create / ptrt_create_interleaved with the device of each) and handing them
 * to a farm, which gathers their RGB8 images onto the device of the FIRST context -- the presenting device, whose
 * viewer maps the frame (rtgl::map_pbo_device_ptr).  Contexts on the presenting device are copied device-to-device;
 * contexts on other devices send with ncclSend / ncclRecv over RCCL (xGMI), one grouped call per frame; strips are
 * scattered to their places.  No host synchronisation inside a frame (out_is_device != 0); a context's next render
 * waits on its own stream until its image has been taken.  The contexts stay the caller's (uploads, options, destroy
 * them AFTER the farm); they must tile the frame exactly once, else PTRT_E_INVALID.
 *   ptrt_farm_render  = ptrt_render(ctx, frame, spp, depth, NULL, 0) on every context + ptrt_farm_gather
 *   ptrt_farm_gather  = gather of the images the contexts hold (for callers that render through the Scene mirror)
 *   ptrt_farm_transport: "device-copy" (all contexts on one device), "rccl", or "peer-copy": hipMemcpyPeerAsync of a remote
 *                     context's image onto the presenting device, on the farm's stream behind the context's render -- no RCCL
 *                     involved (SURVEY 8(e) names it as the alternative).  "peer-copy" is what a farm uses when librccl cannot be
 *                     loaded or ncclCommInitAll fails, with PTRT_FARM_TRANSPORT=peer in the environment, or after
 *                     ptrt_farm_set_option(farm, "transport", 1) (0 = back to RCCL, refused if its communicators never came
 *                     up).  Errors: ptrt_last_error(NULL).
 * The "rccl" transport is UNVERIFIED ON HARDWARE (every test so far ran on a one-GPU box); "peer-copy" has run with equal source
 * and destination device only (PTRT_FARM_FORCE_REMOTE=1: a test hook that makes a farm treat every part but the first as remote).  A presentation-ring slot as the
 * device target (rtgl::map_pbo_device_ptr) is recognised: its download is ordered behind the gather's copies. */
typedef struct ptrt_farm ptrt_farm;
int ptrt_farm_create(ptrt_ctx *const *contexts, int n_contexts, ptrt_farm **out);
int ptrt_farm_bands(const ptrt_farm *farm);
const char *ptrt_farm_transport(const ptrt_farm *farm);
/* ABI 6: 1 if part `part` sits on the presenting device (it may render straight into the frame the gather assembles:
 * ptrt_render(..., frame, PTRT_OUT_DEVICE_FRAME)), 0 if its image has to travel (render it with a NULL target) */
int ptrt_farm_part_is_local(const ptrt_farm *farm, int part);
int ptrt_farm_render(ptrt_farm *farm, int frame_index, int spp, int max_depth, void *out_rgb8, int out_is_device);
int ptrt_farm_gather(ptrt_farm *farm, void *out_rgb8, int out_is_device);
/* ABI 5.  The per-part host work of a frame runs on one worker thread per context (a ptrt_render costs 20-50 us of host
 * time; eight in a row would be the same order as an eighth of a frame on the GPU):
 *   ptrt_farm_parallel   fn(i, user) for every context i at once, each on its own thread (the caller's takes part 0); returns
 *                        when all are back; fn must not throw.  ptrt_farm_render uses it for its ptrt_render calls, the
 *                        C++ TileFarm for its Scene::render_to_device calls.
 *   ptrt_farm_host_us    host time (us) of the caller's thread inside the last ptrt_farm_render
 *   ptrt_farm_device_frame  the device frame a gather into (out_rgb8, out_is_device) assembles: out_rgb8 itself, or the
 *                        farm's own frame when the target is host memory.  A context on the presenting device that rendered
 *                        straight into it (ptrt_render(..., frame, PTRT_OUT_DEVICE_FRAME), as ptrt_farm_render makes them do)
 *                        is not copied by the gather, only waited for
 *   ptrt_farm_set_option "parallel" 0|1 (default 1), "spin_us" (a worker polls that long for the next frame before it
 *                        sleeps; default 2000), "transport" 0|1 (see ptrt_farm_transport) */
int ptrt_farm_parallel(ptrt_farm *farm, void (*fn)(int part, void *user), void *user);
double ptrt_farm_host_us(const ptrt_farm *farm);
void *ptrt_farm_device_frame(ptrt_farm *farm, void *out_rgb8, int out_is_device);
int ptrt_farm_set_option(ptrt_farm *farm, const char *name, long long value);
int ptrt_farm_sync(ptrt_farm *farm);
void ptrt_farm_destroy(ptrt_farm *farm);

/* tuning / diagnostics knobs, by name; unknown names return PTRT_E_INVALID.  None changes a bit of
 * any output (tests force every value and compare with the oracle):
 *   count_rays 0|1        maintain the ptrt_get_stats counters
 *   force_geom -1..2      force a more general traversal variant       force_full 0|1  all material branches
 *   pair_trace 0|1        (ray, mesh) pair compaction                  pair_split 0|1  lanes per pair in partial batches
 *   fetch_min 0..64       idle lanes before the pair queue refills     leaf_pairs 0|1  compacted leaf phase
 *   leaf_min 1..64        lanes waiting at a leaf that end the descent steal 0..64     shadow-ray subtree stealing
 *   csteal 0..64          PMODE 2 closest hit: verified subtree stealing (2).  Once the pair queue is empty an idle lane takes the
 *                         BOTTOM entry of a busy walk's stack -- what that walk would visit last -- with a copy of its ray and limit,
 *                         walks it and merges what it finds; the node loop yields every `csteal` steps while lanes idle.  A thief's hit
 *                         closer than its own leaf box's entry, or two walks of one (ray, mesh) pair reporting the same distance, mark
 *                         the ray, which is then traced again without stealing: same bits.  csteal_min: node steps a
 *                         walk must have taken before it is stolen from (0); csteal_follow 0|1: a thief keeps taking its victim's
 *                         limit (1); csteal_leaf_min 1..64: leaf_min of a stealing phase (32).  0: off (showcase 3.60 -> 3.33 ms).
 *   lds_nodes 0|1         PMODE 2 in 256-thread workgroups sharing an LDS copy of the BLAS top levels
 *   merged -1|0|1         one traversal per loop iteration: a light sample's shadow ray rides with the next extension ray;
 *                         -1 (default): both shapes (equal bit for bit) take turns over a scene's frames 4-9, three samples
 *                         each; the merged one stays only if its median kernel time is at least 0.5 % below the separate-phase
 *                         default's.  The decision, at frame 10, is the ONE place where the otherwise asynchronous ptrt_render
 *                         blocks the host (a hipEventSynchronize on frame 9, once per scene and setting); set merged to 0 or 1
 *                         before the first frame to opt out.  Never taken while the stream is being captured into a hipGraph.
 *   stage 0..7            PMODE 1: shading inputs kept in LDS (0 none; else jitter inputs, |1 light records, |2 material records)
 *   pipeline 0|1, split 1..4   frame pipelining (default 1, 2).  A frame whose launch need not wait for the stream is dealt, by rows
 *                         of 8x8 tiles, to `split` launches on auxiliary streams of the context; launch i follows launch i of the
 *                         previous frame (the same pixels) and the frame is joined onto the context's stream by events, so
 *                         WHAT FOLLOWS ptrt_render ON THE STREAM STILL FOLLOWS THE FRAME -- but the frame itself may start while
 *                         the previous frame's last waves drain and while what the caller enqueued on the stream AFTER the
 *                         previous ptrt_render call still runs; it waits for everything that was on the stream when that
 *                         previous call was made, i.e. for the consumers of every frame but the last
 *                         (Cornell 1080p 1.81 -> 1.68 ms, showcase 3.85 -> 3.57).  Only when nothing else can have a claim on
 *                         what the frame reads or overwrites: a DEVICE target other than the previous frame's (double buffering:
 *                         whatever consumes the previous target on the stream is still entitled to it), no reduced render
 *                         size (with the denoiser or bloom the context keeps two sets of HDR image and G-buffers and
 *                         alternates, so that a frame's post chain and the next frame's trace do not share one), no ptrt_* call since the previous ptrt_render other than the host-only ones
 *                         (ptrt_set_camera / _sky / _option / _prev_view_proj, ptrt_get_option, ptrt_sync, ptrt_last_error,
 *                         ptrt_set_bloom(0)), no pointer from ptrt_device_buffer in the caller's hands, not while the loop shape is
 *                         being sampled or the stream captured.  Any other frame is ONE launch
 *                         ordered behind the stream, as with pipeline = 0.  ptrt_get_option "pipelined" says which the last one was.
 *   refill 0|1|2, persist N   PMODE 1, lane refill: the launch is a grid of persistent waves (`persist` per CU; 0 = what the
 *                         kernel's occupancy holds) that draw the launch's 8x8 tiles from a queue, and a lane whose pixel is finished
 *                         takes the next pixel instead of idling until the slowest pixel of its tile is done (18 % of the
 *                         lane-iterations of a 1080p Cornell frame); the image is tonemapped by a pass behind the launch.  Which lane
 *                         renders a pixel changes nothing in it: same bits.  1 (default): where it was measured to pay -- frames that
 *                         overlap their predecessor ("pipeline"), simple materials, no post chain, spp x bounces >= 16, at least two
 *                         tiles per persistent wave (1080p 4 spp 4 bounces 1.67 -> 1.63 ms, 8 bounces 2.12 -> 1.89, 4K 6.65 -> 6.27);
 *                         2: wherever PMODE 1 runs; 0: never.  ptrt_get_option "refilled" says what the last frame did.
 *                         ticket_tiles 1..16: consecutive tiles per draw from the queue (1; more only pays where the counter
 *                         itself binds -- 1 spp: 0.61 -> 0.49 ms with 4, still behind the one-tile-per-wave kernel's 0.43).
 *   sample_sync -1|0|1    the lanes of a wave start their samples together (1) instead of each as soon as its path has ended (0):
 *                         the wave's lanes then sit at the same bounce, the light-sample phases are skipped by the whole wave at a
 *                         first hit and the primary rays are made once per sample for 64 lanes; lanes wait for the sample's
 *                         longest path.  Same bits.  -1 (default): on where it was measured to pay -- at most 4 bounces (Cornell
 *                         1.76 -> 1.64 ms, 136 meshes 15.9 -> 14.8, the fluid frame -3 %, scenes of short paths even; 5 bounces and more lose).
 *                         ptrt_get_option "sample_sync_eff" says what the last frame did.
 *   tile_run 0..64        the one-tile-per-workgroup kernels' workgroup -> tile map.  Consecutive workgroups go to the eight XCDs in
 *                         turn; with n > 0, of every 8 n consecutive tiles XCD x renders tiles [x n, (x + 1) n) -- neighbours,
 *                         whose rays walk the same part of the trees, share an L2 -- instead of every eighth tile.  8 (default):
 *                         the fluid frame 0.941 -> 0.913 ms with a quarter of its L2 misses, showcase 3.18 -> 3.14, at 4K 24.5 -> 24.1
 *                         (4 .. 32 are within one percent of each other from 720p to 4K); 0: tile k on workgroup k.
 *                         Same bits (a permutation of independent tiles).
 *   tlas_rounds 0|1       real TLAS: shadow rays take one TLAS leaf per fill of the pair list (what > 1024 meshes use) instead of all
 *   pm1_wg 0|1|2          PMODE 1: tiles per workgroup (1 default; 2: two tiles share the LDS copies, six waves per SIMD; 0: 2 if it fits)
 *   lds_pad 0..32768      spare bytes of LDS per workgroup: fewer waves per CU (A/B of the occupancy)
 *   wavefront 0|1, async_lanes 0|1, shade_min 1..64   the alternative loop shapes of DESIGN.md 3.9
 *   wf_sort 0|1|2|4       wavefront stages: the shade stage bins 1 / 2 / 4 groups of 256 consecutive paths by class (finished,
 *                         regenerating, miss, hit mesh x specular flag x roulette) in LDS before shading them -- active-path
 *                         sorting; same bits; measured slower than unsorted, default 0
 *   time_kernels 0|1      1 (default): two events around the trace kernel feed ptrt_kernel_ms_history / ptrt_last_kernel_ms
 *   time_launches 0|1     1: three events around every launch of a frame that is dealt to the auxiliary streams ("pipeline"), on the
 *                         stream the launch runs on: ptrt_launch_ms_history (default 0: six more driver calls per frame)
 *   tm_prio 0..3          lane refill's tonemap pass: |1 on a stream of the highest priority (hipStreamCreateWithPriority) forked from
 *                         and joined to its launch's stream, |2 its waves at s_setprio 3 (A/B: DESIGN.md 3.12)
 *   atrous_exp 0|1        denoiser: 1 = the a-trous luminance weight exp(-dl^2 / 2 sigma^2) through the hardware exponential (v_exp_f32), as
 *                         the reference's `__expf` (denoiser.cuh:731); 0 (default) = the deterministic exponential the oracle defines,
 *                         bit-exact.  Mode 1 is held to a stated tolerance against the oracle (tests/test_denoiser.py: per-pixel
 *                         relative L2 of the denoised HDR image <= 1e-5 (measured 3.9e-7), RGB8 within 1 LSB)
 *   denoiser_active, motion_vectors, use_graphs 0|1 */
int ptrt_set_option(ptrt_ctx *ctx, const char *name, long long value);
/* Reads an option back, and -- read-only -- what the last ptrt_render launched, so that a measurement can name the kernel it
 * timed: render_mode (0 megakernel, 1 wavefront stages, 2 asynchronous lanes), pmode (0 lock-step, 1 pairs over LDS-staged
 * triangles, 2 pair queue, 3 TLAS rounds, 4 merged queue), merged_eff (loop shape of that launch), merged_decided (0 while
 * merged = -1 is still sampling), launches.  ABI 5. */
int ptrt_get_option(ptrt_ctx *ctx, const char *name, long long *value);

/* Render on a caller-owned HIP stream (a `hipStream_t` passed as void*; NULL returns to the
 * context's own stream).  Lets a host that already orders work on a stream (a GL-interop map,
 * an RCCL gather) enqueue frames without host synchronisation, like the reference's use of
 * the default stream (SURVEY 8(b) "Threading / streams"). */
int ptrt_set_stream(ptrt_ctx *ctx, void *hip_stream);

/* Durations (ms) of the most recent path-trace kernel launches, oldest first, measured with
 * HIP events on the stream the kernel ran on; at most 256 are kept.  Synchronises.  Returns
 * the number written (<= max_n) or a negative error. */
int ptrt_kernel_ms_history(ptrt_ctx *ctx, float *out_ms, int max_n);

/* Frames that overlap on the device (option "pipeline") run as `split` launches on auxiliary streams; the events of
 * ptrt_kernel_ms_history then bracket the frame INTERVAL on the context's stream.  With option "time_launches" on, this returns
 * the durations of the launches themselves, oldest first, over the unbroken run of such frames that ends with the last
 * ptrt_render: trace_ms[k] = the path-trace kernel of launch k (HIP events on the stream it ran on), tail_ms[k] (may be NULL) =
 * from its end to the end of the tonemap pass that follows it with lane refill (0 otherwise).  A frame contributes `split`
 * entries.  Synchronises.  Returns the number written (<= max_n) or a negative error.  No counterpart in the reference (it
 * times nothing); what bench.py's `roofline.kernel_ms` is measured with.  ABI 6. */
int ptrt_launch_ms_history(ptrt_ctx *ctx, float *trace_ms, float *tail_ms, int max_n);

/* duration in milliseconds of the last ptrt_render's path-trace kernel and
 * tonemap kernel, measured with HIP events on the context's stream
 * (synchronises).  Either pointer may be NULL. */
int ptrt_last_kernel_ms(ptrt_ctx *ctx, float *trace_ms, float *tonemap_ms);

#ifdef __cplusplus
}
#endif
#endif /* PTRT_H */

// ptrt_host_capi.cpp -- flat C entry points over the C++ `Scene` mirror
// (host/ptrt/scene.hpp) so that Python (ctypes) tests, bench.py and other FFI
// callers can drive the same host API a C++ application uses.  One function per
// Scene method; C++ exceptions become negative return codes + hs_last_error().
#include "../host/ptrt/scene.hpp"
#include "../host/ptrt/serialize.hpp"
#include "../host/ptrt/farm.hpp"
#include "../host/ptrt/view.hpp"

#include <chrono>
#include <cstring>
#include <string>

namespace {
thread_local std::string g_err;

Material from_floats(const float *m) { // 27 floats, field order of hs_material_default
    Material r;
    r.albedo = vec3(m[0], m[1], m[2]);
    r.specular = vec3(m[3], m[4], m[5]);
    r.metallic = m[6];
    r.roughness = m[7];
    r.emission = vec3(m[8], m[9], m[10]);
    r.ior = m[11];
    r.transmission = m[12];
    r.transmissionRoughness = m[13];
    r.clearcoat = m[14];
    r.clearcoatRoughness = m[15];
    r.subsurfaceColor = vec3(m[16], m[17], m[18]);
    r.subsurfaceRadius = m[19];
    r.anisotropy = m[20];
    r.sheen = m[21];
    r.sheenTint = vec3(m[22], m[23], m[24]);
    r.iridescence = m[25];
    r.iridescenceThickness = m[26];
    return r;
}
void to_floats(const Material &r, float *m) {
    m[0] = r.albedo.x; m[1] = r.albedo.y; m[2] = r.albedo.z;
    m[3] = r.specular.x; m[4] = r.specular.y; m[5] = r.specular.z;
    m[6] = r.metallic; m[7] = r.roughness;
    m[8] = r.emission.x; m[9] = r.emission.y; m[10] = r.emission.z;
    m[11] = r.ior; m[12] = r.transmission; m[13] = r.transmissionRoughness;
    m[14] = r.clearcoat; m[15] = r.clearcoatRoughness;
    m[16] = r.subsurfaceColor.x; m[17] = r.subsurfaceColor.y; m[18] = r.subsurfaceColor.z;
    m[19] = r.subsurfaceRadius; m[20] = r.anisotropy; m[21] = r.sheen;
    m[22] = r.sheenTint.x; m[23] = r.sheenTint.y; m[24] = r.sheenTint.z;
    m[25] = r.iridescence; m[26] = r.iridescenceThickness;
}
int mesh_index(Scene *s, Mesh *m) {
    for (size_t i = 0; i < s->getMeshCount(); ++i)
        if (s->getMesh(i) == m)
            return (int)i;
    return -1;
}
} // namespace

#define HS_TRY(body)                                                                                            \
    try {                                                                                                       \
        body;                                                                                                   \
    } catch (const std::exception &e) {                                                                         \
        g_err = e.what();                                                                                       \
        return -1;                                                                                              \
    }

extern "C" {

const char *hs_last_error(void) { return g_err.c_str(); }

void hs_material_default(float *out27) { to_floats(Material(), out27); }
// Material(albedo, roughness, metallic) constructor (material_lib.cuh:91-104)
void hs_material_make(const float *albedo3, float roughness, float metallic, float *out27) {
    to_floats(Material(vec3(albedo3[0], albedo3[1], albedo3[2]), roughness, metallic), out27);
}

void *hs_scene_create(int w, int h, int tile_y0, int tile_rows, int device) {
    try {
        return new Scene(w, h, tile_y0, tile_rows, device);
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}
void *hs_scene_create_interleaved(int w, int h, int phase, int period, int device) {
    try {
        return new Scene(w, h, Scene::Interleave{phase, period}, device);
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}
int hs_tile_rows(void *s) { return static_cast<Scene *>(s)->getTileRows(); }
void hs_scene_destroy(void *s) { delete static_cast<Scene *>(s); }
void *hs_backend(void *s) { return static_cast<Scene *>(s)->backend(); }

int hs_init_blue_noise(void *s) { HS_TRY(static_cast<Scene *>(s)->initBlueNoise()); return 0; }
void hs_blue_noise_table(float *out8192) {
    const std::vector<float> &t = ptrtBlueNoiseTable();
    std::memcpy(out8192, t.data(), t.size() * sizeof(float));
}
// raw generator, for tests of the relaxation itself at small sizes
void hs_blue_noise_generate(int size, int iterations, float *out) {
    std::vector<float> t = BlueNoiseGenerator::generateBlueNoise2D(size, iterations);
    std::memcpy(out, t.data(), t.size() * sizeof(float));
}

int hs_add_cube(void *s, const float *mat27) {
    Scene *sc = static_cast<Scene *>(s);
    HS_TRY(return mesh_index(sc, sc->addCube(from_floats(mat27))));
}
int hs_add_sphere(void *s, int segments, const float *mat27) {
    Scene *sc = static_cast<Scene *>(s);
    HS_TRY(return mesh_index(sc, sc->addSphere(segments, from_floats(mat27))));
}
int hs_add_plane_xz(void *s, float y, float half, const float *mat27) {
    Scene *sc = static_cast<Scene *>(s);
    HS_TRY(return mesh_index(sc, sc->addPlaneXZ(y, half, from_floats(mat27))));
}
int hs_add_triangles(void *s, const float *verts9, int n_tris, const float *mat27) {
    Scene *sc = static_cast<Scene *>(s);
    std::vector<Triangle> tris;
    tris.reserve(n_tris);
    for (int i = 0; i < n_tris; ++i) {
        const float *v = verts9 + (size_t)i * 9;
        tris.emplace_back(vec3(v[0], v[1], v[2]), vec3(v[3], v[4], v[5]), vec3(v[6], v[7], v[8]));
    }
    HS_TRY(return mesh_index(sc, sc->addTriangles(tris, from_floats(mat27))));
}
int hs_add_mesh_obj(void *s, const char *path, const float *mat27) {
    Scene *sc = static_cast<Scene *>(s);
    HS_TRY(return mesh_index(sc, sc->addMesh(path, from_floats(mat27))));
}
int hs_add_checkerboard(void *s, float y, int tiles, float tile_size, const float *white27, const float *black27) {
    HS_TRY(static_cast<Scene *>(s)->addCheckerboardPlaneXZ(y, tiles, tile_size, from_floats(white27),
                                                            from_floats(black27)));
    return 0;
}

// op: 0 scale(vec3) 1 translate 2 moveTo 3 rotateSelfEulerXYZ 4 setPosition 5 setRotation 6 transform.setScale
int hs_mesh_op(void *s, int mesh, int op, float x, float y, float z) {
    Mesh *m = static_cast<Scene *>(s)->getMesh((size_t)mesh);
    if (!m) {
        g_err = "no such mesh";
        return -1;
    }
    const vec3 v(x, y, z);
    switch (op) {
    case 0: m->scale(v); break;
    case 1: m->translate(v); break;
    case 2: m->moveTo(v); break;
    case 3: m->rotateSelfEulerXYZ(v); break;
    case 4: m->setPosition(v); break;
    case 5: m->setRotation(v); break;
    case 6: m->transform.setScale(v); m->transform.updateMatrices(); break;
    default: g_err = "unknown mesh op"; return -1;
    }
    return 0;
}
// replaces the vertex array of a mesh in place (the dynamic-geometry caller,
// PTRTtransfer.cuh:2249-2270: new positions, same topology)
int hs_mesh_set_vertices(void *s, int mesh, const float *xyz, int n_verts) {
    Mesh *m = static_cast<Scene *>(s)->getMesh((size_t)mesh);
    if (!m || (size_t)n_verts != m->vertices.size()) {
        g_err = "vertex count mismatch";
        return -1;
    }
    for (int i = 0; i < n_verts; ++i)
        m->vertices[i] = vec3(xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]);
    m->bvhDirty = true;
    m->vertsDirty = true;
    return 0;
}
// What updatePTScene does to a `Triangles` mesh whose triangleVerts changed (PTRTtransfer.cuh:2249-2270): vertices and faces
// rewritten as an unshared-vertex soup, both dirty flags set, the local box recomputed -- the caller commits afterwards.
static double g_soup_us = 0.0; // host time of hs_mesh_set_triangle_soup so far: the CALLER's share of a commit (bench.py --via-commit)
int hs_commit_host_us(void *s, double *caller_us, double *mirror_us, double *compare_us) {
    const Scene *sc = static_cast<Scene *>(s);
    if (caller_us)
        *caller_us = g_soup_us;
    if (mirror_us)
        *mirror_us = sc->commitHostMicros();
    if (compare_us)
        *compare_us = sc->commitCompareMicros();
    return 0;
}
int hs_mesh_set_triangle_soup(void *s, int mesh, const float *verts9, int n_tris) {
    struct Timer {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~Timer() { g_soup_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
    } timer;
    Mesh *m = static_cast<Scene *>(s)->getMesh((size_t)mesh);
    if (!m || !verts9 || n_tris < 1) {
        g_err = "hs_mesh_set_triangle_soup: bad argument";
        return -1;
    }
    m->vertices.clear();
    m->faces.clear();
    m->vertices.reserve((size_t)n_tris * 3);
    m->faces.reserve((size_t)n_tris);
    for (int i = 0; i < n_tris; ++i) {
        const int base = (int)m->vertices.size();
        for (int k = 0; k < 3; ++k)
            m->vertices.push_back(vec3(verts9[i * 9 + k * 3], verts9[i * 9 + k * 3 + 1], verts9[i * 9 + k * 3 + 2]));
        m->faces.push_back(Tri{base, base + 1, base + 2});
    }
    m->bvhDirty = true;
    m->vertsDirty = true;
    m->computeLocalAABB();
    return 0;
}
int hs_set_dynamic_geometry_policy(void *s, int policy) {
    if (policy < 0 || policy > 2) {
        g_err = "policy must be 0 (HostRebuild), 1 (GpuRefit) or 2 (GpuRebuild)";
        return -1;
    }
    static_cast<Scene *>(s)->setDynamicGeometryPolicy((Scene::DynamicGeometryPolicy)policy);
    return 0;
}
int hs_commit_counts(void *s, long long *gpu_commits, long long *geometry_uploads) {
    const Scene *sc = static_cast<Scene *>(s);
    if (gpu_commits)
        *gpu_commits = (long long)sc->gpuDynamicCommitCount();
    if (geometry_uploads)
        *geometry_uploads = (long long)sc->geometryUploadCount();
    return 0;
}
int hs_mesh_counts(void *s, int mesh, int *n_verts, int *n_faces, int *n_nodes) {
    Mesh *m = static_cast<Scene *>(s)->getMesh((size_t)mesh);
    if (!m)
        return -1;
    *n_verts = (int)m->vertices.size();
    *n_faces = (int)m->faces.size();
    *n_nodes = (int)m->bvhNodes.size();
    return 0;
}

void hs_add_point_light(void *s, const float *pos3, const float *col3, float intensity, float range, float radius) {
    static_cast<Scene *>(s)->addPointLight(vec3(pos3[0], pos3[1], pos3[2]), vec3(col3[0], col3[1], col3[2]), intensity,
                                           range, radius);
}
void hs_add_directional_light(void *s, const float *dir3, const float *col3, float intensity) {
    static_cast<Scene *>(s)->addDirectionalLight(vec3(dir3[0], dir3[1], dir3[2]), vec3(col3[0], col3[1], col3[2]),
                                                 intensity);
}
void hs_add_spot_light(void *s, const float *pos3, const float *dir3, const float *col3, float intensity, float inner,
                       float outer, float range, float radius) {
    static_cast<Scene *>(s)->addSpotLight(vec3(pos3[0], pos3[1], pos3[2]), vec3(dir3[0], dir3[1], dir3[2]),
                                          vec3(col3[0], col3[1], col3[2]), intensity, inner, outer, range, radius);
}
void hs_move_light_to(void *s, int i, const float *pos3) {
    static_cast<Scene *>(s)->moveLightTo((size_t)i, vec3(pos3[0], pos3[1], pos3[2]));
}

void hs_set_camera(void *s, const float *from3, const float *at3, const float *up3, float vfov, float aperture,
                   float focus_dist) {
    static_cast<Scene *>(s)->setCamera(vec3(from3[0], from3[1], from3[2]), vec3(at3[0], at3[1], at3[2]),
                                       vec3(up3[0], up3[1], up3[2]), vfov, aperture, focus_dist);
}
void hs_move_camera(void *s, const float *pos3) { static_cast<Scene *>(s)->moveCamera(vec3(pos3[0], pos3[1], pos3[2])); }
void hs_look_camera_at(void *s, const float *at3) {
    static_cast<Scene *>(s)->lookCameraAt(vec3(at3[0], at3[1], at3[2]));
}
void hs_set_sky_gradient(void *s, const float *top3, const float *bottom3) {
    static_cast<Scene *>(s)->setSkyGradient(vec3(top3[0], top3[1], top3[2]), vec3(bottom3[0], bottom3[1], bottom3[2]));
}
void hs_disable_sky(void *s) { static_cast<Scene *>(s)->disableSky(); }
int hs_load_hdri(void *s, const char *path) { HS_TRY(static_cast<Scene *>(s)->loadHDRI(path)); return 0; }
int hs_set_environment_map(void *s, const float *rgba, int w, int h) {
    HS_TRY(static_cast<Scene *>(s)->setEnvironmentMap(rgba, w, h));
    return 0;
}
void hs_free_hdri(void *s) { static_cast<Scene *>(s)->freeHDRI(); }

void hs_set_bvh_leaf_target(void *s, int target, int tol) { static_cast<Scene *>(s)->setBVHLeafTarget(target, tol); }
void hs_set_max_bounce_depth(void *s, int d) { static_cast<Scene *>(s)->setMaxBounceDepth(d); }
void hs_set_samples_per_pixel(void *s, int spp) { static_cast<Scene *>(s)->setSamplesPerPixel(spp); }
void hs_set_perf_samples_per_pixel(void *s, int spp) { static_cast<Scene *>(s)->setPerfSamplesPerPixel(spp); }
void hs_set_max_depth(void *s, int d) { static_cast<Scene *>(s)->setMaxDepth(d); }
int hs_get_samples_per_pixel(void *s) { return static_cast<Scene *>(s)->getSamplesPerPixel(); }
int hs_set_denoiser_enabled(void *s, int e) { HS_TRY(static_cast<Scene *>(s)->setDenoiserEnabled(e != 0)); return 0; }
void hs_set_bloom_enabled(void *s, int e) { static_cast<Scene *>(s)->setBloomEnabled(e != 0); }
int hs_set_performance_preset(void *s, const char *name) { HS_TRY(static_cast<Scene *>(s)->setPerformancePreset(name)); return 0; }
int hs_set_resolution_scale(void *s, float scale) { HS_TRY(static_cast<Scene *>(s)->setResolutionScale(scale)); return 0; }
void hs_get_render_size(void *s, int *w, int *h) {
    *w = static_cast<Scene *>(s)->getRenderWidth();
    *h = static_cast<Scene *>(s)->getRenderHeight();
}
void hs_get_settings(void *s, int *spp, int *depth, int *denoiser, int *bloom, float *scale) {
    const Scene::PerformanceSettings &p = static_cast<Scene *>(s)->getPerformanceSettings();
    *spp = p.samplesPerPixel;
    *depth = p.maxBounceDepth;
    *denoiser = p.enableDenoiser;
    *bloom = p.enableBloom;
    *scale = p.resolutionScale;
}
int hs_set_mesh_material(void *s, int mesh, const float *mat27) {
    Scene *sc = static_cast<Scene *>(s);
    sc->setMeshMaterial((size_t)mesh, from_floats(mat27));
    sc->commitMaterialChanges();
    return 0;
}

// which: 0 = proj*view of the previous frame (what the next render's motion pass uses), 1 = current
void hs_get_view_proj(void *s, int which, float *out16) {
    const mat4 m = which ? static_cast<Scene *>(s)->getViewProjMatrix() : static_cast<Scene *>(s)->getPrevViewProjMatrix();
    std::memcpy(out16, m.m, sizeof m.m);
}
int hs_upload(void *s) { HS_TRY(static_cast<Scene *>(s)->uploadToGPU()); return 0; }
int hs_commit_object_changes(void *s) { HS_TRY(static_cast<Scene *>(s)->commitObjectChanges()); return 0; }
int hs_refit_object_changes(void *s) { HS_TRY(static_cast<Scene *>(s)->refitObjectChanges()); return 0; }
int hs_refit_from_device(void *s, int mesh, const void *device_xyz) {
    HS_TRY(static_cast<Scene *>(s)->refitFromDevice((size_t)mesh, static_cast<const float *>(device_xyz)));
    return 0;
}
int hs_refit_from_host(void *s, int mesh, const float *host_xyz) {
    HS_TRY(static_cast<Scene *>(s)->refitFromHost((size_t)mesh, host_xyz));
    return 0;
}
int hs_rebuild_object_changes(void *s, int sync_host_copy) {
    HS_TRY(static_cast<Scene *>(s)->rebuildObjectChanges(sync_host_copy != 0));
    return 0;
}
int hs_rebuild_from_device(void *s, int mesh, const void *device_xyz) {
    HS_TRY(static_cast<Scene *>(s)->rebuildFromDevice((size_t)mesh, static_cast<const float *>(device_xyz)));
    return 0;
}
int hs_update_triangles(void *s, int mesh, const void *verts9, int tri_count, int on_device) {
    HS_TRY(static_cast<Scene *>(s)->updateTriangles((size_t)mesh, static_cast<const float *>(verts9), tri_count, on_device != 0));
    return 0;
}
int hs_mesh_prim_indices(void *s, int mesh, int *out, int count) {
    const Mesh *m = static_cast<Scene *>(s)->getMesh((size_t)mesh);
    if (!m || count != (int)m->bvhPrimIndices.size())
        return -1;
    std::memcpy(out, m->bvhPrimIndices.data(), (size_t)count * sizeof(int));
    return 0;
}
int hs_render_to_device(void *s, void *device_pixels) {
    HS_TRY(static_cast<Scene *>(s)->render_to_device(static_cast<unsigned char *>(device_pixels)));
    return 0;
}
int hs_render_to_host(void *s, void *host_pixels) {
    HS_TRY(static_cast<Scene *>(s)->render_to_host(static_cast<unsigned char *>(host_pixels)));
    return 0;
}
int hs_post_frame(void *s, const float *accum, const float *normal, const float *depth, const int *object_id,
                  void *pixels, int is_device) {
    HS_TRY(static_cast<Scene *>(s)->postFrameFromDevice(accum, normal, depth, object_id,
                                                        static_cast<unsigned char *>(pixels), is_device));
    return 0;
}
int hs_get_frame_count(void *s) { return static_cast<Scene *>(s)->getFrameCount(); }
void hs_set_frame_count(void *s, int f) { static_cast<Scene *>(s)->setFrameCount(f); }
int hs_trace_single_ray(void *s, const float *o3, const float *d3, ptrt_hit *out) {
    HitInfo h = static_cast<Scene *>(s)->traceSingleRay(vec3(o3[0], o3[1], o3[2]), vec3(d3[0], d3[1], d3[2]));
    out->hit = h.hit;
    out->t = h.t;
    out->point = {h.point.x, h.point.y, h.point.z};
    out->normal = {h.normal.x, h.normal.y, h.normal.z};
    out->mesh_index = h.mesh_index;
    out->front_face = h.front_face;
    out->u = h.u;
    out->v = h.v;
    out->face_index = h.face_index;
    out->local_point = {h.localPoint.x, h.localPoint.y, h.localPoint.z};
    return 0;
}
int hs_save_ppm(void *s, const char *path, unsigned char *pixels) {
    HS_TRY(static_cast<Scene *>(s)->saveAsPPM(path, pixels));
    return 0;
}

// The reference's frame loop (readme.txt:100-119 / glfw_view_interop.hpp:281-332) over host/ptrt/view.hpp:
// renders `frames` frames through the presentation ring; every presented frame is copied to
// out_frames (frames * W*H*3 bytes, may be NULL) in presentation order; returns wall ms per frame
// (PCIe-inclusive) in *ms_per_frame.  dump_prefix non-empty: every dump_every-th frame as PPM.
int hs_view_run(void *s, int frames, int slots, unsigned char *out_frames, double *ms_per_frame, const char *dump_prefix,
                int dump_every) {
    try {
        Scene &scene = *static_cast<Scene *>(s);
        rtgl::InteropViewer V;
        rtgl::init_interop_viewer(V, scene, "headless", slots);
        V.dump_prefix = dump_prefix ? dump_prefix : "";
        V.dump_every = dump_every;
        const size_t bytes = (size_t)scene.getWidth() * scene.getTileRows() * 3; // a band context presents its rows
        size_t got = 0;
        auto present = [&](bool flush) {
            rtgl::blit_pbo_to_texture(V, flush);
            if (V.host_frame) {
                if (out_frames)
                    std::memcpy(out_frames + got * bytes, V.host_frame, bytes);
                ++got;
                rtgl::draw_interop(V);
            }
        };
        const auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; ++f) {
            uint8_t *d = rtgl::map_pbo_device_ptr(V);
            scene.render_to_device(d);
            rtgl::unmap_pbo(V);
            present(false);
        }
        while (got < (size_t)frames)
            present(true); // drain the ring
        const auto t1 = std::chrono::steady_clock::now();
        if (ms_per_frame)
            *ms_per_frame = std::chrono::duration<double, std::milli>(t1 - t0).count() / (frames > 0 ? frames : 1);
        rtgl::destroy_interop_viewer(V);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// flattened scene (host pointers owned by the Scene, valid until it is mutated)
const ptrt_scene_desc *hs_flatten(void *s) {
    try {
        return &static_cast<Scene *>(s)->flatten();
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}


// C++ TileFarm (host/ptrt/farm.hpp) from Python: `n` parts on `devices`, scene built by `recipe` (0 Cornell-like
// cubes are built by the caller through hs_farm_scene); returns the farm or NULL
void *hs_farm_create(int w, int h, const int *devices, int n, int strips) {
    try {
        return new TileFarm(w, h, std::vector<int>(devices, devices + n), strips ? TileFarm::Strips : TileFarm::Bands,
                            [](Scene &) {});
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}
void *hs_farm_scene(void *f, int i) { return &static_cast<TileFarm *>(f)->scene((size_t)i); }
int hs_farm_size(void *f) { return (int)static_cast<TileFarm *>(f)->size(); }
const char *hs_farm_transport(void *f) { return static_cast<TileFarm *>(f)->transport(); }
int hs_farm_render(void *f, unsigned char *pixels, int is_device) {
    HS_TRY(is_device ? static_cast<TileFarm *>(f)->render_to_device(pixels) : static_cast<TileFarm *>(f)->render_to_host(pixels));
    return 0;
}
double hs_farm_host_us(void *f) { return static_cast<TileFarm *>(f)->hostMicroseconds(); }
int hs_farm_set_parallel(void *f, int on) { HS_TRY(static_cast<TileFarm *>(f)->setParallel(on != 0)); return 0; }
int hs_farm_sync(void *f) { HS_TRY(static_cast<TileFarm *>(f)->sync()); return 0; }
void hs_farm_destroy(void *f) { delete static_cast<TileFarm *>(f); }

// canonical byte stream of the flattened scene (host/ptrt/serialize.hpp); returns its length, copies
// min(length, cap) bytes
size_t hs_serialize_scene(void *s, unsigned char *out, size_t cap) {
    try {
        const std::vector<uint8_t> b = ptrt_detail::serialize_scene(static_cast<Scene *>(s)->flatten());
        if (out && cap)
            std::memcpy(out, b.data(), b.size() < cap ? b.size() : cap);
        return b.size();
    } catch (const std::exception &e) {
        g_err = e.what();
        return 0;
    }
}

} // extern "C"

// ptrt/scene.hpp -- the host API of the PTRT path tracer, over the MI355X back end.
//
// `class Scene` keeps the reference's method names, defaults and error behaviour
// (reference: src/pathtracer/scene/scene.cuh:78-2001) so a caller of the CUDA
// renderer can switch headers.  What differs is underneath: the scene is
// flattened once per change into plain arrays and handed to the C ABI of
// include/ptrt.h, which owns every device allocation.  Out of scope here (see
// DESIGN.md): denoiser, bloom, resolution scaling, HDRI sky, wireframe and
// debug-visualisation helpers -- their setters are accepted and recorded so call
// sites compile, and `render_to_device` reports when one is enabled.
#pragma once
#include "hdr.hpp"
#include "mesh.hpp"

#include <chrono>
#include <cstdint>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <utility>
#include <vector>

// ---- Material (scene/material_lib.cuh:12-105) -----------------------------------------
struct Material {
    vec3 albedo{0.8f}, specular{0.04f};
    float metallic = 0.0f, roughness = 0.5f;
    vec3 emission{0.0f};
    float ior = 1.5f, transmission = 0.0f, transmissionRoughness = 0.0f;
    float clearcoat = 0.0f, clearcoatRoughness = 0.03f;
    vec3 subsurfaceColor{1.0f};
    float subsurfaceRadius = 0.0f, anisotropy = 0.0f, sheen = 0.0f;
    vec3 sheenTint{0.5f};
    float iridescence = 0.0f, iridescenceThickness = 550.0f;
    Material() = default;
    Material(const vec3 &alb, float rough = 0.5f, float met = 0.0f) {
        albedo = alb;
        roughness = rough;
        metallic = met;
        specular = lerp(vec3(0.04f), albedo, metallic);
        transmissionRoughness = fmaxf(transmissionRoughness, roughness);
    }
};

// conversion helpers of material_lib.cuh:128-146, used by the applications' material libraries (app_utils.cuh:79-103)
inline float phongShininessToRoughness(float n) {
    const float alpha = sqrtf(2.0f / (fmaxf(n, 1.0f) + 2.0f));
    return fminf(fmaxf(fmaxf(alpha, 0.02f), 0.0f), 1.0f);
}
inline float iorToF0(float ior) {
    const float a = (ior - 1.0f) / (ior + 1.0f);
    return a * a;
}

// ---- Light (scene/lights.cuh) ---------------------------------------------------------
enum LightType { LIGHT_POINT = 0, LIGHT_DIRECTIONAL = 1, LIGHT_SPOT = 2 };
struct Light {
    LightType type = LIGHT_POINT;
    vec3 position{0, 10, 0}, direction{0, -1, 0}, color{1.0f};
    float intensity = 1.0f, range = 100.0f, innerCone = 0.5f, outerCone = 0.7f, radius = 0.0f;
};

// ---- Camera (scene/camera.cuh:32-205, ray-generation state only) ----------------------
class Camera {
    vec3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    float lens_radius = 0.0f, fov = 90.0f, aspect = 1.0f, near_clip = 0.1f, far_clip = 1000.0f;
    mat4 view_matrix, proj_matrix; // for motion vectors (camera.cuh:41-43,88-95)

    void update_matrices(const vec3 &lookfrom, const vec3 &lookat, const vec3 &vup) {
        view_matrix = mat4::lookAt(lookfrom, lookat, vup);
        proj_matrix = mat4::perspective(fov * (PI / 180.0f), aspect, near_clip, far_clip);
    }
    void rebuild(const vec3 &lookat, const vec3 &vup, float focus_dist) {
        update_matrices(origin, lookat, vup);
        w = (origin - lookat).normalized();
        u = cross(vup, w).normalized();
        v = cross(w, u);
        const float theta = fov * (PI / 180.0f);
        const float h = tanf(theta / 2.0f);
        const float viewport_height = 2.0f * h;
        const float viewport_width = aspect * viewport_height;
        horizontal = focus_dist * viewport_width * u;
        vertical = focus_dist * viewport_height * v;
        lower_left_corner = origin - horizontal * 0.5f - vertical * 0.5f - focus_dist * w;
    }

  public:
    Camera(vec3 lookfrom, vec3 lookat, vec3 vup, float vfov, float aspect_ratio, float aperture = 0.0f,
           float focus_dist = 1.0f, float znear = 0.1f, float zfar = 1000.0f) {
        origin = lookfrom;
        fov = vfov;
        aspect = aspect_ratio;
        near_clip = znear;
        far_clip = zfar;
        rebuild(lookat, vup, focus_dist);
        lens_radius = aperture / 2.0f;
    }
    explicit Camera(float aspect_ratio, float viewport_height = 2.0f, float focal_length = 1.0f) {
        origin = vec3(0.0f);
        const float viewport_width = viewport_height * aspect_ratio;
        horizontal = vec3(viewport_width, 0.0f, 0.0f);
        vertical = vec3(0.0f, viewport_height, 0.0f);
        lower_left_corner = origin - horizontal * 0.5f - vertical * 0.5f - vec3(0.0f, 0.0f, focal_length);
        u = vec3(1, 0, 0);
        v = vec3(0, 1, 0);
        w = vec3(0, 0, 1);
        aspect = aspect_ratio;
        fov = 90.0f;
        near_clip = 0.1f;
        far_clip = 100.0f;
        update_matrices(origin, vec3(0, 0, -1), vec3(0, 1, 0));
    }
    mat4 get_view_proj() const { return proj_matrix * view_matrix; } // camera.cuh:257-259
    // get_ray_simple (camera.cuh:201-205); the debug-ray helper is the only host caller
    void get_ray(float s, float t, vec3 &o, vec3 &d) const {
        o = origin;
        d = (lower_left_corner + s * horizontal + t * vertical - origin).normalized();
    }
    vec3 get_origin() const { return origin; }
    vec3 get_lower_left_corner() const { return lower_left_corner; }
    vec3 get_horizontal() const { return horizontal; }
    vec3 get_vertical() const { return vertical; }
    void set_position(const vec3 &pos) { // camera.cuh:271-299
        vec3 old_center = lower_left_corner + 0.5f * horizontal + 0.5f * vertical;
        float focus_dist = (origin - old_center).length();
        vec3 lookat = origin - w * focus_dist;
        vec3 vup = v;
        origin = pos;
        rebuild(lookat, vup, (origin - lookat).length());
    }
    void look_at(const vec3 &target, const vec3 &vup = vec3(0, 1, 0)) { // camera.cuh:306-331
        rebuild(target, vup, (origin - target).length());
    }
    ptrt_camera flat() const {
        auto c = [](const vec3 &a) { return ptrt_vec3{a.x, a.y, a.z}; };
        return ptrt_camera{c(origin), c(lower_left_corner), c(horizontal), c(vertical), c(u), c(v), c(w), lens_radius};
    }
};

// ---- blue-noise table (common/bluenoise.cuh:79-198) -----------------------------------
// 64x64 jittered-stratified points relaxed by 25 rounds of O(N^2) toroidal
// repulsion.  std::uniform_real_distribution<float> is implementation-defined;
// the table this produces with libstdc++ is pinned in tests/golden/.
struct BlueNoiseGenerator {
    static std::vector<float> generateBlueNoise2D(int size, int relaxation_iterations) {
        const int n = size * size;
        std::vector<float> px(n), py(n), fx(n), fy(n);
        std::mt19937 rng(12345);
        std::uniform_real_distribution<float> dist(0.0f, 1.0f);
        const float cell = 1.0f / size;
        for (int y = 0; y < size; ++y)
            for (int x = 0; x < size; ++x) {
                const float jx = dist(rng); // x first, then y, one pair per point
                const float jy = dist(rng);
                px[y * size + x] = (x + jx) * cell;
                py[y * size + x] = (y + jy) * cell;
            }
        const float step = 0.0001f, min_d2 = 0.0001f;
        auto wrap = [](float a) {
            float r = std::fmod(a, 1.0f);
            return r < 0.0f ? r + 1.0f : r;
        };
        for (int it = 0; it < relaxation_iterations; ++it) {
            for (int i = 0; i < n; ++i) {
                float ax = 0.0f, ay = 0.0f;
                const float xi = px[i], yi = py[i];
                for (int j = 0; j < n; ++j) {
                    if (i == j)
                        continue;
                    float dx = xi - px[j], dy = yi - py[j];
                    if (dx > 0.5f) dx -= 1.0f;
                    if (dx < -0.5f) dx += 1.0f;
                    if (dy > 0.5f) dy -= 1.0f;
                    if (dy < -0.5f) dy += 1.0f;
                    float d2 = dx * dx + dy * dy;
                    d2 = std::max(d2, min_d2);
                    const float inv = 1.0f / d2;
                    ax += dx * inv;
                    ay += dy * inv;
                }
                fx[i] = ax;
                fy[i] = ay;
            }
            for (int i = 0; i < n; ++i) {
                const float mag = std::sqrt(fx[i] * fx[i] + fy[i] * fy[i]);
                if (mag < 1e-6f)
                    continue;
                px[i] = wrap(px[i] + (fx[i] / mag) * step);
                py[i] = wrap(py[i] + (fy[i] / mag) * step);
            }
        }
        std::vector<float> out((size_t)n * 2);
        for (int i = 0; i < n; ++i) {
            out[i * 2] = px[i];
            out[i * 2 + 1] = py[i];
        }
        return out;
    }
};
// the table initBlueNoise() would upload (bluenoise.cuh:189-198), computed once per process
inline const std::vector<float> &ptrtBlueNoiseTable() {
    static const std::vector<float> t = BlueNoiseGenerator::generateBlueNoise2D(PTRT_BLUE_NOISE_SIZE, 25);
    return t;
}

struct HitInfo { // math/intersection.cuh:108-124
    bool hit = false;
    float t = 1e30f;
    vec3 point, normal;
    int mesh_index = -1;
    bool front_face = true;
    float u = 0, v = 0;
    int face_index = -1;
    vec3 localPoint;
};

// ---- Scene ----------------------------------------------------------------------------
// device of `Scene(w, h)`; -1 = host-only scenes (build / flatten / inspect on a box without a GPU)
#ifndef PTRT_DEFAULT_DEVICE
#define PTRT_DEFAULT_DEVICE 0
#endif
class Scene {
  public:
    struct PerformanceSettings { // scene.cuh:189-199
        bool enableDenoiser = true;
        bool enableBloom = true;
        bool enableMotionVectors = true;
        int maxBounceDepth = 4;
        int samplesPerPixel = 1;
        float resolutionScale = 1.0f;
        bool fastBVHUpdates = true;          // never read by the reference either
        bool enableRussianRoulette = true;   // never read: RR always starts at bounce 2
        int russianRouletteStartBounce = 1;  // never read (path_logic.cuh:24)
    };

    // Scene(w,h) of the reference; the extra arguments select a band of rows and a
    // device for tile-parallel rendering and default to "whole frame,
    // device 0".
    Scene(int w, int h, int tile_y0 = 0, int tile_rows = 0, int device = PTRT_DEFAULT_DEVICE)
        : width(w), height(h), camera(static_cast<float>(w) / h, 2.0f, 1.0f) {
        tileRows = tile_rows > 0 ? tile_rows : h;
        tileY0 = tile_y0;
        device_ = device;
        render_width = w;
        render_height = h;
        if (device < 0)
            return; // host-only scene: build/flatten/inspect, no back end (every GPU call then fails loudly)
        int rc = ptrt_create(w, h, tile_y0, tile_rows, device, &ctx);
        if (rc != PTRT_OK) {
            std::string msg = std::string("Failed to create GPU context: ") + ptrt_last_error(ctx);
            ptrt_destroy(ctx);
            ctx = nullptr;
            throw std::runtime_error(msg);
        }
        try { // a constructor that throws never runs ~Scene: the context must not outlive it
            check(ptrt_reset_rng(ctx, PTRT_DEFAULT_SEED), "Failed to init rand states"); // scene.cuh:433-456
            if (tileRows == h && tile_y0 == 0) {
                // `denoiser_ = new Denoiser(settings)` (scene.cuh:1984-1993): exists from construction,
                // used while perfSettings.enableDenoiser; band contexts cannot denoise (filters cross bands)
                fullFrame = true;
                check(ptrt_denoiser_enable(ctx, nullptr), "Failed to create denoiser");
                denoiserAllocated = true;
            }
        } catch (...) {
            ptrt_destroy(ctx);
            ctx = nullptr;
            throw;
        }
        prev_view_proj = camera.get_view_proj();
    }
    // Tile farm, second way of cutting the frame: this scene renders every `period`-th 8-row strip starting with
    // strip `phase` (ptrt_create_interleaved).  No post chain, like a band.
    struct Interleave {
        int phase, period;
    };
    Scene(int w, int h, Interleave il, int device = PTRT_DEFAULT_DEVICE)
        : width(w), height(h), camera(static_cast<float>(w) / h, 2.0f, 1.0f) {
        tileY0 = il.phase * 8;
        device_ = device;
        render_width = w;
        render_height = h;
        tileRows = 0;
        for (int t = il.phase; t * 8 < h; t += (il.period < 1 ? 1 : il.period))
            tileRows += (t * 8 + 8 <= h) ? 8 : h - t * 8;
        if (device < 0)
            return;
        if (ptrt_create_interleaved(w, h, il.phase, il.period, device, &ctx) != PTRT_OK) {
            std::string msg = std::string("Failed to create GPU context: ") + ptrt_last_error(ctx);
            ptrt_destroy(ctx);
            ctx = nullptr;
            throw std::runtime_error(msg);
        }
        try {
            check(ptrt_reset_rng(ctx, PTRT_DEFAULT_SEED), "Failed to init rand states");
        } catch (...) {
            ptrt_destroy(ctx);
            ctx = nullptr;
            throw;
        }
        prev_view_proj = camera.get_view_proj();
    }
    ~Scene() { ptrt_destroy(ctx); }
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;

    // the application's `initBlueNoise()` call; without it the table is all zeros,
    // exactly as in the reference (bluenoise.cuh:46,189)
    void initBlueNoise() { needBackend(); check(ptrt_set_blue_noise(ctx, ptrtBlueNoiseTable().data()), "blue noise upload failed"); }
    void setBlueNoiseTable(const float *table) { needBackend(); check(ptrt_set_blue_noise(ctx, table), "blue noise upload failed"); }

    // ---- accumulation / acceleration settings (scene.cuh:1270-1296) -----------------
    void resetAccumulation() {
        frame_count_ = 0;
        prev_view_proj = camera.get_view_proj(); // scene.cuh:1282: no motion across a reset
    }
    void setBVHLeafTarget(int target, int tol = 5) {
        bvhLeafTarget_ = target < 1 ? 1 : target;
        bvhLeafTol_ = tol < 0 ? 0 : tol;
        for (auto &m : meshes)
            m->bvhDirty = true;
        resetAccumulation();
    }

    // ---- camera (scene.cuh:1298-1331) -----------------------------------------------
    void setCamera(const vec3 &lookfrom, const vec3 &lookat, const vec3 &vup, float vfov, float aperture = 0.0f,
                   float focus_dist = 1.0f) {
        camera = Camera(lookfrom, lookat, vup, vfov, static_cast<float>(width) / height, aperture, focus_dist);
        cameraDirty = true;
        resetAccumulation();
    }
    void setCameraSimple(float viewport_height = 2.0f, float focal_length = 1.0f) {
        camera = Camera(static_cast<float>(width) / height, viewport_height, focal_length);
        cameraDirty = true;
        resetAccumulation();
    }
    vec3 cameraOrigin() const { return camera.get_origin(); }
    vec3 cameraForward() const {
        return (camera.get_lower_left_corner() + camera.get_horizontal() * 0.5f + camera.get_vertical() * 0.5f -
                camera.get_origin())
            .normalized();
    }
    void moveCamera(const vec3 &pos) { camera.set_position(pos); cameraDirty = true; resetAccumulation(); }
    void lookCameraAt(const vec3 &target, const vec3 &vup = vec3(0, 1, 0)) {
        camera.look_at(target, vup);
        cameraDirty = true;
        resetAccumulation();
    }
    Camera &getCamera() { cameraDirty = true; return camera; }

    // ---- geometry (scene.cuh:1334-1500) ---------------------------------------------
    Mesh *addMesh(const std::string &obj_path, const Material &mat = Material()) {
        return push(std::make_unique<Mesh>(obj_path), mat);
    }
    Mesh *addTriangles(const std::vector<Triangle> &tris, const Material &mat = Material()) {
        auto m = std::make_unique<Mesh>();
        m->vertices.clear();
        m->faces.clear();
        m->vertices.reserve(tris.size() * 3);
        m->faces.reserve(tris.size());
        for (const Triangle &t : tris) {
            const int base = (int)m->vertices.size();
            m->vertices.push_back(t.v0);
            m->vertices.push_back(t.v1);
            m->vertices.push_back(t.v2);
            m->faces.push_back(Tri{base, base + 1, base + 2});
        }
        return push(std::move(m), mat);
    }
    Mesh *addPlaneXZ(float planeY, float halfSize, const Material &mat = Material(vec3(0.8f))) {
        const vec3 A(-halfSize, planeY, -halfSize), B(halfSize, planeY, -halfSize), C(halfSize, planeY, halfSize),
            D(-halfSize, planeY, halfSize);
        return addTriangles({Triangle(A, C, B), Triangle(A, D, C)}, mat); // CCW from +Y
    }
    void addCheckerboardPlaneXZ(float planeY, int tilesPerSide, float tileSize, const Material &whiteMat,
                                const Material &blackMat) {
        std::vector<Triangle> white, black;
        const int N = tilesPerSide;
        const float start = -N * tileSize;
        for (int iz = 0; iz < 2 * N; ++iz)
            for (int ix = 0; ix < 2 * N; ++ix) {
                const float x0 = start + ix * tileSize, x1 = x0 + tileSize;
                const float z0 = start + iz * tileSize, z1 = z0 + tileSize;
                const vec3 A(x0, planeY, z0), B(x1, planeY, z0), C(x1, planeY, z1), D(x0, planeY, z1);
                auto &bucket = (((ix + iz) & 1) == 0) ? white : black;
                bucket.emplace_back(A, C, B);
                bucket.emplace_back(A, D, C);
            }
        if (!white.empty())
            addTriangles(white, whiteMat);
        if (!black.empty())
            addTriangles(black, blackMat);
    }
    Mesh *addCube(const Material &mat = Material(vec3(1.0f, 0.0f, 0.0f))) { return push(std::make_unique<Mesh>(), mat); }
    // UV sphere of diameter 1: (segments+1)^2 vertices, 2*segments^2 faces (scene.cuh:1455-1500)
    Mesh *addSphere(int segments = 32, const Material &mat = Material(vec3(1.0f, 0.0f, 0.0f))) {
        auto m = std::make_unique<Mesh>();
        m->vertices.clear();
        m->faces.clear();
        const int rings = segments, sectors = segments;
        const float radius = 0.5f;
        for (int r = 0; r <= rings; ++r) {
            const float phi = PI * float(r) / float(rings);
            const float y = cosf(phi) * radius;
            const float ringRadius = sinf(phi) * radius;
            for (int s = 0; s <= sectors; ++s) {
                const float theta = TWO_PI * float(s) / float(sectors);
                m->vertices.push_back(vec3(ringRadius * cosf(theta), y, ringRadius * sinf(theta)));
            }
        }
        for (int r = 0; r < rings; ++r)
            for (int s = 0; s < sectors; ++s) {
                const int curr = r * (sectors + 1) + s, next = curr + sectors + 1;
                m->faces.push_back({curr, next, curr + 1});
                m->faces.push_back({curr + 1, next, next + 1});
            }
        return push(std::move(m), mat);
    }

    // ---- lights (scene.cuh:1503-1545, 1800-1827) -------------------------------------
    void addPointLight(const vec3 &position, const vec3 &color, float intensity = 1.0f, float range = 100.0f,
                       float radius = 0.0f) {
        Light l;
        l.type = LIGHT_POINT;
        l.position = position;
        l.color = color;
        l.intensity = intensity;
        l.range = range;
        l.radius = radius;
        addLight(l);
    }
    void addDirectionalLight(const vec3 &direction, const vec3 &color, float intensity = 1.0f) {
        Light l;
        l.type = LIGHT_DIRECTIONAL;
        l.direction = direction.normalized();
        l.color = color;
        l.intensity = intensity;
        addLight(l);
    }
    // cone angles are given in radians and stored as cosines
    void addSpotLight(const vec3 &position, const vec3 &direction, const vec3 &color, float intensity = 1.0f,
                      float innerCone = 0.5f, float outerCone = 0.7f, float range = 100.0f, float radius = 0.0f) {
        Light l;
        l.type = LIGHT_SPOT;
        l.position = position;
        l.direction = direction.normalized();
        l.color = color;
        l.intensity = intensity;
        l.innerCone = cosf(innerCone);
        l.outerCone = cosf(outerCone);
        l.range = range;
        l.radius = radius;
        addLight(l);
    }
    Light *getLight(size_t i) { return i < lights.size() ? &lights[i] : nullptr; }
    size_t getLightCount() const { return lights.size(); }
    void moveLightTo(size_t i, const vec3 &position) {
        if (Light *l = getLight(i)) {
            l->position = position;
            lightsDirty = true;
            resetAccumulation();
        }
    }
    void commitLightChanges() {
        if (!lights.empty()) {
            lightsDirty = true;
            resetAccumulation();
        }
    }

    // ---- sky (scene.cuh:1548-1565) ---------------------------------------------------
    void setSkyGradient(const vec3 &top, const vec3 &bottom) {
        sky_color_top = top;
        sky_color_bottom = bottom;
        use_sky = true;
        skyDirty = true;
        resetAccumulation();
    }
    void disableSky() { use_sky = false; skyDirty = true; resetAccumulation(); }
    // scene.cuh:958-1026: an equirectangular .hdr picture becomes the sky (stbi_loadf, 4 channels,
    // flipped vertically -> host/ptrt/hdr.hpp reads the same floats); sampleSky then looks it up
    // instead of the gradient.  setEnvironmentMap is the same without the file.
    void loadHDRI(const std::string &filepath) {
        std::cout << "Loading HDRI: " << filepath << "..." << std::endl;
        std::vector<float> rgba;
        int w = 0, h = 0;
        ptrt_detail::load_radiance_hdr(filepath, w, h, rgba);
        std::cout << "Loaded HDRI: " << w << "x" << h << std::endl;
        setEnvironmentMap(rgba.data(), w, h);
    }
    void setEnvironmentMap(const float *rgba, int w, int h) {
        if (!rgba || w < 1 || h < 1)
            throw std::runtime_error("setEnvironmentMap: empty map");
        envMap.assign(rgba, rgba + (size_t)w * h * 4);
        env_width = w;
        env_height = h;
        envDirty = true;
        use_sky = true;
        skyDirty = true;
        resetAccumulation();
    }
    void freeHDRI() { // scene.cuh:944-956
        envMap.clear();
        env_width = env_height = 0;
        envDirty = true;
    }

    // ---- materials / objects ---------------------------------------------------------
    void setMeshMaterial(size_t index, const Material &mat) {
        if (index < mesh_materials.size())
            mesh_materials[index] = mat;
    }
    void commitMaterialChanges() { materialsDirty = true; resetAccumulation(); }
    Mesh *getMesh(size_t i) { return i < meshes.size() ? meshes[i].get() : nullptr; }
    const Mesh *getMesh(size_t i) const { return i < meshes.size() ? meshes[i].get() : nullptr; }
    size_t getMeshCount() const { return meshes.size(); }
    void moveMeshTo(size_t i, const vec3 &p) { if (Mesh *m = getMesh(i)) m->moveTo(p); }
    void translateMesh(size_t i, const vec3 &d) { if (Mesh *m = getMesh(i)) m->translate(d); }
    void rotateMesh(size_t i, const vec3 &r) { if (Mesh *m = getMesh(i)) m->rotateSelfEulerXYZ(r); }
    void scaleMesh(size_t i, float s) { if (Mesh *m = getMesh(i)) m->scale(s); }
    void scaleMesh(size_t i, const vec3 &s) { if (Mesh *m = getMesh(i)) m->scale(s); }
    void commitObjectChanges() { updateAccelerationStructures(); resetAccumulation(); } // scene.cuh:1784
    // What commitObjectChanges() / render_to_device() do with a mesh whose vertices were rewritten (bvhDirty / vertsDirty set
    // by the caller, as updatePTScene does for a `Triangles` mesh, PTRTtransfer.cuh:2249-2270, or by the vertex-baking
    // transforms) while its vertex count and face list are what was uploaded:
    //   HostRebuild (default) -- the reference: Mesh::buildBVH on the CPU, every mesh re-laid-out and re-uploaded
    //                            (scene.cuh:656-733);
    //   GpuRefit   -- the uploaded tree is kept: the new positions go to the device (ptrt_update_vertices) and the boxes are
    //                 refitted there (ptrt_refit); no host build, no re-upload of the arena;
    //   GpuRebuild -- as GpuRefit, but the faces are re-assigned to the leaves in Morton order first (ptrt_build_bvh).
    // A changed face list, vertex count, a new mesh, or a mesh without an uploaded tree falls back to HostRebuild for that
    // commit.  The host copy of a tree kept on the GPU is brought up to date when something reads it (flatten(), a real
    // TLAS, the next full upload).  So the UNCHANGED call sequence updatePTScene(scene, unified) -> commitObjectChanges()
    // reaches the GPU refit once the application has said  scene.setDynamicGeometryPolicy(...)  after building it.
    enum class DynamicGeometryPolicy { HostRebuild = 0, GpuRefit = 1, GpuRebuild = 2 };
    void setDynamicGeometryPolicy(DynamicGeometryPolicy p) {
        // (chosen after the upload: the face lists of the meshes nobody has touched since are what the device holds)
        if (p != DynamicGeometryPolicy::HostRebuild && gpu_resources_initialized && !geometryDirty && uploadedFaces.size() == meshes.size())
            for (size_t i = 0; i < meshes.size(); ++i)
                if (uploadedFaces[i].empty() && !meshes[i]->bvhDirty && !meshes[i]->vertsDirty)
                    uploadedFaces[i] = meshes[i]->faces;
        dynPolicy = p;
    }
    DynamicGeometryPolicy getDynamicGeometryPolicy() const { return dynPolicy; }
    // commits that took the GPU path / that re-uploaded the geometry (tests, bench)
    // host time (microseconds, accumulated) the mirror itself spends in a commit -- updateAccelerationStructures(): the
    // dynamic-geometry policy's topology check, the vertex hand-over, the refit / rebuild launches, the host structures --
    // and of that the part spent comparing face lists; for measurements (bench.py --via-commit), not part of the reference's API
    double commitHostMicros() const { return commitMicros; }
    double commitCompareMicros() const { return compareMicros; }
    size_t gpuDynamicCommitCount() const { return gpuDynamicCommits; }
    size_t geometryUploadCount() const { return geometryUploads; }
    // Dynamic geometry with unchanged topology (the fluid-sim caller, PTRTtransfer.cuh:2249-2270):
    // instead of commitObjectChanges()' full CPU rebuild, keep every tree and refit its boxes --
    // on the host copy (so flatten() stays consistent) and on the GPU (ptrt_update_vertices +
    // ptrt_refit, no re-upload of the arena).  Not in the reference.
    void refitObjectChanges() {
        needBackend();
        if (!gpu_resources_initialized || geometryDirty)
            throw std::runtime_error("refitObjectChanges: call uploadToGPU() first (topology must already be on the GPU)");
        for (size_t i = 0; i < meshes.size(); ++i) {
            Mesh *m = meshes[i].get();
            if (!m->vertsDirty)
                continue;
            m->refitBVH();
            m->vertsDirty = false;
            check(ptrt_update_vertices(ctx, (int)i, &m->vertices[0].x, (int)m->vertices.size(), 0),
                  "Failed to update vertices");
        }
        check(ptrt_refit(ctx), "Failed to refit");
        syncTLAS();
        resetAccumulation();
    }
    // TLAS over the meshes' new boxes: the host copy always; the device copy too when the TLAS has inner nodes
    // (the GPU refit moves the root box of a single-leaf TLAS itself and leaves a real TLAS to the host, which
    // rebuilds it as the reference's commit does)
    void syncTLAS() {
        buildTLAS();
        flat.tlas_nodes = h_tlasNodes.data();
        flat.tlas_node_count = (int)h_tlasNodes.size();
        flat.tlas_mesh_indices = h_tlasMeshIndices.data();
        flat.tlas_index_count = (int)h_tlasMeshIndices.size();
        if (h_tlasNodes.size() > 1)
            check(ptrt_update_instances(ctx, flat.meshes, flat.mesh_count, flat.tlas_nodes, flat.tlas_node_count,
                                        flat.tlas_mesh_indices, flat.tlas_index_count),
                  "Failed to update the TLAS");
    }
    // same, the new positions already being in device memory (no host copy is kept: flatten()
    // then describes the LAST host-side vertices)
    void refitFromDevice(size_t mesh, const float *device_xyz) {
        needBackend();
        Mesh *m = getMesh(mesh);
        if (!m)
            throw std::runtime_error("refitFromDevice: no such mesh");
        if (h_tlasNodes.size() > 1)
            throw std::runtime_error("refitFromDevice: a TLAS with inner nodes is rebuilt on the host, which needs the "
                                     "vertices (use setVertices + refitObjectChanges)");
        check(ptrt_update_vertices(ctx, (int)mesh, device_xyz, (int)m->vertices.size(), 1), "Failed to update vertices");
        check(ptrt_refit(ctx), "Failed to refit");
        resetAccumulation();
    }

    // same from HOST memory without touching the host copy of the mesh (its vertices are whatever the caller last put
    // there; the host tree is refitted when something reads it): ptrt_update_vertices(on_device = 0) + ptrt_refit
    void refitFromHost(size_t mesh, const float *host_xyz) {
        needBackend();
        Mesh *m = getMesh(mesh);
        if (!m)
            throw std::runtime_error("refitFromHost: no such mesh");
        if (h_tlasNodes.size() > 1)
            throw std::runtime_error("refitFromHost: a TLAS with inner nodes is rebuilt on the host (use setVertices + refitObjectChanges)");
        check(ptrt_update_vertices(ctx, (int)mesh, host_xyz, (int)m->vertices.size(), 0), "Failed to update vertices");
        check(ptrt_refit(ctx), "Failed to refit");
        resetAccumulation();
    }

    // commitObjectChanges() for meshes whose face count is unchanged, with the BVH rebuilt ON THE GPU
    // (ptrt_build_bvh: Morton-order median split over the uploaded tree shape + refit) instead of
    // Mesh::buildBVH on the CPU + re-upload (scene.cuh:656-733).  The host copy of each rebuilt tree
    // follows (prim order read back, boxes refitted) so flatten() describes what the GPU traverses.
    void rebuildObjectChanges(bool syncHostCopy = true) {
        needBackend();
        if (!gpu_resources_initialized || geometryDirty)
            throw std::runtime_error("rebuildObjectChanges: call uploadToGPU() first");
        for (size_t i = 0; i < meshes.size(); ++i) {
            Mesh *m = meshes[i].get();
            if (!m->vertsDirty)
                continue;
            m->vertsDirty = false;
            check(ptrt_update_vertices(ctx, (int)i, &m->vertices[0].x, (int)m->vertices.size(), 0),
                  "Failed to update vertices");
            check(ptrt_build_bvh(ctx, (int)i), "Failed to build BVH");
            if (syncHostCopy)
                syncPrimOrder(i);
        }
        if (syncHostCopy)
            syncTLAS();
        else if (h_tlasNodes.size() > 1)
            throw std::runtime_error("rebuildObjectChanges: a TLAS with inner nodes needs the host copy (syncHostCopy)");
        resetAccumulation();
    }
    // same, new positions already in device memory (host copy not updated)
    void rebuildFromDevice(size_t mesh, const float *device_xyz) {
        needBackend();
        Mesh *m = getMesh(mesh);
        if (!m)
            throw std::runtime_error("rebuildFromDevice: no such mesh");
        if (h_tlasNodes.size() > 1)
            throw std::runtime_error("rebuildFromDevice: a TLAS with inner nodes is rebuilt on the host, which needs the "
                                     "vertices (use setVertices + rebuildObjectChanges)");
        check(ptrt_update_vertices(ctx, (int)mesh, device_xyz, (int)m->vertices.size(), 1), "Failed to update vertices");
        check(ptrt_build_bvh(ctx, (int)mesh), "Failed to build BVH");
        resetAccumulation();
    }
    // updatePTScene's `Triangles` path (PTRTtransfer.cuh:2204-2385) for a soup mesh made by
    // addTriangles with room for its original triangle count: `tri_count` new triangles (9 floats
    // each) from host or device memory, BVH rebuilt on the GPU.  The host copy of the mesh gets the
    // same padded vertices when the data is on the host.
    void updateTriangles(size_t mesh, const float *verts9, int tri_count, bool on_device = false) {
        needBackend();
        Mesh *m = getMesh(mesh);
        if (!m)
            throw std::runtime_error("updateTriangles: no such mesh");
        check(ptrt_update_triangles(ctx, (int)mesh, verts9, tri_count, on_device ? 1 : 0), "Failed to update triangles");
        check(ptrt_build_bvh(ctx, (int)mesh), "Failed to build BVH");
        if (!on_device) {
            const size_t real = (size_t)tri_count * 3;
            for (size_t v = 0; v < m->vertices.size(); ++v) {
                const size_t s = v < real ? v : (real ? real - 1 : 0);
                m->vertices[v] = real ? vec3(verts9[s * 3], verts9[s * 3 + 1], verts9[s * 3 + 2]) : vec3(0.0f);
            }
            syncPrimOrder(mesh);
            syncTLAS();
        }
        resetAccumulation();
    }
    // host copy of mesh `i`'s tree := what the GPU now traverses
    void syncPrimOrder(size_t i) {
        Mesh *m = getMesh(i);
        if (!m)
            throw std::runtime_error("syncPrimOrder: no such mesh");
        check(ptrt_read_prim_order(ctx, (int)i, m->bvhPrimIndices.data(), (int)m->bvhPrimIndices.size()),
              "Failed to read the prim order");
        m->refitBVH();
    }

    bool hasObjectChanges() const {
        for (auto &m : meshes)
            if (m->bvhDirty)
                return true;
        return false;
    }

    // ---- quality knobs (scene.cuh:1833-1912) -----------------------------------------
    void setPerformancePreset(const std::string &p) {
        auto set = [&](bool dn, bool bl, bool mv, int depth, float scale, int rr) {
            perfSettings.enableDenoiser = dn;
            perfSettings.enableBloom = bl;
            perfSettings.enableMotionVectors = mv;
            perfSettings.maxBounceDepth = depth;
            perfSettings.resolutionScale = scale;
            perfSettings.russianRouletteStartBounce = rr;
        };
        if (p == "ultra") { set(false, true, true, 32, 1.0f, 8); perfSettings.samplesPerPixel = 128; }
        else if (p == "quality") set(true, true, true, 6, 1.0f, 2);
        else if (p == "balanced") set(true, true, true, 4, 1.0f, 1);
        else if (p == "performance") set(true, false, true, 3, 0.75f, 1);
        else if (p == "fast") set(false, false, false, 2, 0.35f, 1);
        updateScaledBuffers();
    }
    void setDenoiserEnabled(bool e) {
        if (perfSettings.enableDenoiser != e) {
            perfSettings.enableDenoiser = e;
            updateScaledBuffers(); // a no-op unless the render size changed, as in the reference
        }
    }
    void setBloomEnabled(bool e) { perfSettings.enableBloom = e; }
    void setMaxBounceDepth(int d) { perfSettings.maxBounceDepth = d < 1 ? 1 : (d > 16 ? 16 : d); }
    void setResolutionScale(float s) {
        s = fmaxf(0.25f, fminf(1.0f, s));
        if (fabsf(s - perfSettings.resolutionScale) > 0.01f) {
            perfSettings.resolutionScale = s;
            updateScaledBuffers();
        }
    }
    int getRenderWidth() const { return render_width; }
    int getRenderHeight() const { return render_height; }
    // scene.cuh:1248-1255: stored, accumulation reset -- and IGNORED by render_to_device, which takes its sample
    // count and depth from perfSettings (scene.cuh:86-87, 1044-1045).  Kept that way so a caller of the
    // reference renders the same frame here.
    void setSamplesPerPixel(int spp) { samples_per_pixel_ = spp; resetAccumulation(); }
    void setMaxDepth(int depth) { max_depth_ = depth; resetAccumulation(); }
    int getSamplesPerPixel() const { return samples_per_pixel_; } // scene.cuh:1720
    // NOT in the reference, whose perfSettings.samplesPerPixel is reachable only through the "ultra" preset
    // (scene.cuh:1839): the sample count render_to_device uses; needed to express the 4-spp configurations.
    void setPerfSamplesPerPixel(int spp) { perfSettings.samplesPerPixel = spp < 1 ? 1 : spp; }
    const PerformanceSettings &getPerformanceSettings() const { return perfSettings; }

    // ---- debug visualisation (scene.cuh:1564-1684, app_utils.cuh:304-368): out of scope -- the
    // switches and ray lists are kept so call sites compile and behave, no helper meshes are generated
    void setWireframeMode(bool e) { wireframe_mode = e; }
    bool isWireframeMode() const { return wireframe_mode; }
    void toggleWireframeMode() { wireframe_mode = !wireframe_mode; }
    void setShowFrustum(bool show) { show_frustum = show; }
    void toggleFrustum() { show_frustum = !show_frustum; }
    void setShowRays(bool show) { show_rays = show; }
    void setRayLength(float length) { ray_length = length; }
    void addDebugRay(const vec3 &origin, const vec3 &direction) { debug_rays.push_back({origin, direction, -1.0f}); }
    void addDebugRayWithLength(const vec3 &origin, const vec3 &direction, float length) {
        debug_rays.push_back({origin, direction.normalized(), length});
    }
    void clearDebugRays() { debug_rays.clear(); }
    size_t getDebugRayCount() const { return debug_rays.size(); }
    void clearVisualizationMeshes() {}
    void updateVisualizationMeshes() {
        if ((show_frustum || (show_rays && !debug_rays.empty())) && !warnedViz) {
            std::cerr << "NOTE: frustum / ray visualisation meshes are not generated by this back end\n";
            warnedViz = true;
        }
        show_frustum = false;
    }
    void generatePrimaryRayVisualization(Camera cam, int numRays = 10) {
        clearDebugRays();
        const int gridSize = (int)sqrt((double)numRays);
        for (int y = 0; y < gridSize; ++y)
            for (int x = 0; x < gridSize; ++x) {
                vec3 o, d;
                cam.get_ray((x + 0.5f) / gridSize, (y + 0.5f) / gridSize, o, d);
                addDebugRay(o, d);
            }
        std::cout << "generated primary rays" << '\n';
    }
    // the wireframe debug kernel is out of scope: says so once and shows the path-traced frame instead
    void render_to_device_wireframe(unsigned char *device_pixels, float /*wireframeThickness*/) {
        if (!warnedViz) {
            std::cerr << "NOTE: wireframe rendering is not part of this back end; rendering the path-traced frame\n";
            warnedViz = true;
        }
        render_to_device(device_pixels);
    }

    // ---- upload + render -------------------------------------------------------------
    void uploadToGPU() { // scene.cuh:1643-1657
        if (meshes.empty()) {
            std::cerr << "Warning: No meshes in scene\n";
            return;
        }
        if (meshes.size() != mesh_materials.size())
            throw std::runtime_error("Mesh count and material count mismatch!");
        needBackend();
        updateAccelerationStructures();
        gpu_resources_initialized = true;
        resetAccumulation();
    }

    // One frame: acceleration-structure update, path trace, tonemap into
    // `device_pixels` (W*rows*3 bytes on the device, bottom-up).  Returns without
    // synchronising, like the reference (scene.cuh:1028-1209).
    void render_to_device(unsigned char *device_pixels) { renderInternal(device_pixels, 1); }
    // same frame into HOST memory (synchronous); the reference has no such call on
    // its PT Scene -- callers there map a GL buffer instead
    void render_to_host(unsigned char *host_pixels) { renderInternal(host_pixels, 0); }
    // a band / strip scene's rows straight into the WHOLE frame (W*H*3 bytes, bottom-up, on this scene's device): what the
    // tile farm does with the parts on the presenting GPU -- no image of their own, nothing to copy (ptrt.h PTRT_OUT_DEVICE_FRAME)
    void render_to_frame(unsigned char *device_frame) { renderInternal(device_frame, PTRT_OUT_DEVICE_FRAME); }
    int deviceIndex() const { return device_; }

    // Tile farm, presenting rank: the post chain of render_to_device over a frame gathered from band contexts
    // (device pointers, top-down; include/ptrt.h ptrt_post_frame).  Same bookkeeping as a rendered frame.
    void postFrameFromDevice(const float *accum, const float *normal, const float *depth, const int *object_id,
                             unsigned char *pixels, int is_device) {
        const bool denoise = perfSettings.enableDenoiser && denoiserAllocated;
        const bool bloom = perfSettings.enableBloom && fullFrame && render_width >= 64 && render_height >= 64;
        if (fullFrame)
            check(ptrt_set_bloom(ctx, bloom ? 1 : 0), "bloom");
        if (denoiserAllocated) {
            check(ptrt_set_option(ctx, "denoiser_active", denoise ? 1 : 0), "denoiser option");
            check(ptrt_set_option(ctx, "motion_vectors", (denoise && perfSettings.enableMotionVectors) ? 1 : 0),
                  "denoiser option");
            check(ptrt_set_prev_view_proj(ctx, prev_view_proj.m), "prev view-proj");
        }
        if (cameraDirty) {
            ptrt_camera c = camera.flat();
            check(ptrt_set_camera(ctx, &c), "Failed to set camera");
            cameraDirty = false;
        }
        check(ptrt_post_frame(ctx, accum, normal, depth, object_id, pixels, is_device), "ptrt_post_frame");
        frame_count_++;
        prev_view_proj = camera.get_view_proj();
    }


    HitInfo traceSingleRay(const vec3 &o, const vec3 &d) { // scene.cuh:1367-1391
        HitInfo h;
        needBackend();
        float of[3] = {o.x, o.y, o.z}, df[3] = {d.x, d.y, d.z};
        ptrt_hit r;
        if (ptrt_trace_rays(ctx, of, df, 1, &r) != PTRT_OK) {
            std::cerr << "traceSingleRay failed: " << ptrt_last_error(ctx) << "\n";
            return h;
        }
        h.hit = r.hit != 0;
        h.t = r.t;
        h.point = vec3(r.point.x, r.point.y, r.point.z);
        h.normal = vec3(r.normal.x, r.normal.y, r.normal.z);
        h.mesh_index = r.mesh_index;
        h.front_face = r.front_face != 0;
        h.u = r.u;
        h.v = r.v;
        h.face_index = r.face_index;
        h.localPoint = vec3(r.local_point.x, r.local_point.y, r.local_point.z);
        return h;
    }

    void saveAsPPM(const std::string &filename, unsigned char *pixels) const { // ASCII P3, scene.cuh:1694-1708
        std::ofstream ofs(filename, std::ios::binary);
        if (!ofs)
            throw std::runtime_error("Cannot open file: " + filename);
        ofs << "P3\n" << width << ' ' << height << "\n255\n";
        for (size_t i = 0, n = (size_t)width * height * 3; i < n; i += 3)
            ofs << int(pixels[i]) << ' ' << int(pixels[i + 1]) << ' ' << int(pixels[i + 2]) << '\n';
    }

    int getWidth() const { return width; }
    int getHeight() const { return height; }
    int getTileRows() const { return tileRows; } // rows this context renders (== height unless it is a band)
    int getTileY0() const { return tileY0; }
    int getDevice() const { return device_; }
    size_t getPixelBufferSize() const { return (size_t)width * height * 3; }
    int getFrameCount() const { return frame_count_; }
    // DEVICE pointers, as in the reference (scene.cuh:1722-1725)
    vec3 *getNoisyColorBuffer() { return (vec3 *)ptrt_device_buffer(ctx, PTRT_BUF_ACCUM); }
    vec3 *getNormalBuffer() { return (vec3 *)ptrt_device_buffer(ctx, PTRT_BUF_NORMAL); }
    float *getDepthBuffer() { return (float *)ptrt_device_buffer(ctx, PTRT_BUF_DEPTH); }
    int *getObjectIdBuffer() { return (int *)ptrt_device_buffer(ctx, PTRT_BUF_OBJECT_ID); }
    float *getMotionVectorBuffer() { return (float *)ptrt_device_buffer(ctx, PTRT_BUF_MOTION); } // float2 per pixel
    vec3 *getDenoisedBuffer() { return (vec3 *)ptrt_device_buffer(ctx, PTRT_BUF_DENOISED); }
    mat4 getPrevViewProjMatrix() const { return prev_view_proj; }
    mat4 getViewProjMatrix() const { return camera.get_view_proj(); }

    // ---- access for tests, tools and the tile farm -----------------------------------
    ptrt_ctx *backend() { return ctx; }
    void setFrameCount(int f) { frame_count_ = f; }
    // Flattened view of the current scene (pointers stay valid until the next
    // mutation of the scene).  Brings BLAS/TLAS up to date first.
    const ptrt_scene_desc &flatten() {
        prepareHostStructures();
        return flat;
    }

  private:
    int width, height, tileRows = 0, tileY0 = 0, device_ = 0;
    int samples_per_pixel_ = 16, max_depth_ = 8; // scene.cuh:86-87: stored, ignored by render_to_device
    int frame_count_ = 0;
    struct DebugRay {
        vec3 origin, direction;
        float length; // < 0: the scene's ray_length
    };
    std::vector<DebugRay> debug_rays;
    bool wireframe_mode = false, show_frustum = false, show_rays = false, warnedViz = false;
    float ray_length = 5.0f;
    int bvhLeafTarget_ = 12, bvhLeafTol_ = 5; // scene.cuh:90-91
    std::vector<std::unique_ptr<Mesh>> meshes;
    std::vector<Material> mesh_materials;
    std::vector<Light> lights;
    Camera camera;
    bool use_sky = true;
    vec3 sky_color_top{0.6f, 0.7f, 1.0f}, sky_color_bottom{1.0f, 1.0f, 1.0f}; // scene.cuh:164-166
    PerformanceSettings perfSettings;
    bool gpu_resources_initialized = false;

    ptrt_ctx *ctx = nullptr;
    bool geometryDirty = true, materialsDirty = true, lightsDirty = true, cameraDirty = true, skyDirty = true;
    bool instancesDirty = false; // only instance transforms (and with them the TLAS) changed since the last upload
    bool warnedPost = false, denoiserAllocated = false;
    std::vector<float> envMap;                // RGBA floats of the HDRI sky (empty: gradient)
    int env_width = 0, env_height = 0;
    bool envDirty = false;
    bool fullFrame = false;                   // band (tile) contexts have no post chain
    int render_width = 0, render_height = 0;  // scene.cuh:203-204
    mat4 prev_view_proj; // proj*view of the previous frame (scene.cuh:113)

    // scene.cuh:1913-2000: the size the path tracer renders at follows perfSettings.resolutionScale
    // (>= 64, and here also <= the frame); when it changes, the low-resolution buffers and the
    // denoiser are re-created (the denoiser only if enabled) and accumulation restarts.
    void updateScaledBuffers() {
        int nw = static_cast<int>(width * perfSettings.resolutionScale);
        int nh = static_cast<int>(height * perfSettings.resolutionScale);
        nw = nw < 64 ? 64 : nw;
        nh = nh < 64 ? 64 : nh;
        nw = nw > width ? width : nw;
        nh = nh > height ? height : nh;
        if (nw == render_width && nh == render_height)
            return;
        if (ctx && !fullFrame)
            return; // a band context always renders its rows of the full-size frame
        render_width = nw;
        render_height = nh;
        if (ctx) {
            check(ptrt_set_render_size(ctx, nw, nh), "Failed to resize the render buffers");
            denoiserAllocated = false; // freed by the resize
            if (perfSettings.enableDenoiser) {
                check(ptrt_denoiser_enable(ctx, nullptr), "Failed to create denoiser");
                denoiserAllocated = true;
            }
        }
        resetAccumulation();
    }

    // flattened arrays handed to the back end
    std::vector<ptrt_mesh_desc> flatMeshes;
    std::vector<mat4> lastWorld;
    std::vector<DeviceBVHNode> h_tlasNodes;
    std::vector<int> h_tlasMeshIndices;
    std::vector<ptrt_light> flatLights;
    struct MatSoA {
        std::vector<ptrt_vec3> albedo, specular, emission, subsurfaceColor, sheenTint;
        std::vector<float> metallic, roughness, ior, transmission, transmissionRoughness, clearcoat,
            clearcoatRoughness, subsurfaceRadius, anisotropy, sheen, iridescence, iridescenceThickness;
    } soa;
    ptrt_scene_desc flat{};
    // dynamic-geometry policy: the face lists and vertex counts of the last upload, and which host trees lag the device's
    DynamicGeometryPolicy dynPolicy = DynamicGeometryPolicy::HostRebuild;
    std::vector<std::vector<Tri>> uploadedFaces;
    std::vector<size_t> uploadedVerts;
    std::vector<unsigned char> hostTreeStale, uploadedSoup;
    size_t gpuDynamicCommits = 0, geometryUploads = 0;
    double commitMicros = 0.0, compareMicros = 0.0;

    void needBackend() const {
        if (!ctx)
            throw std::runtime_error("host-only Scene (device < 0) has no GPU back end");
    }
    void check(int rc, const char *what) {
        if (rc != PTRT_OK)
            throw std::runtime_error(std::string(what) + ": " + ptrt_last_error(ctx));
    }
    Mesh *push(std::unique_ptr<Mesh> m, const Material &mat) {
        meshes.push_back(std::move(m));
        mesh_materials.push_back(mat);
        geometryDirty = materialsDirty = true;
        resetAccumulation();
        return meshes.back().get();
    }
    void addLight(const Light &l) {
        lights.push_back(l);
        lightsDirty = true;
        resetAccumulation();
    }

    // TLAS over the meshes' world AABBs, same builder and leaf limit as the BLAS
    // (scene.cuh:458-594)
    void buildTLAS() {
        std::vector<ptrt_detail::BuildRef> refs(meshes.size());
        for (size_t i = 0; i < meshes.size(); ++i) {
            Mesh *m = meshes[i].get();
            if (m->transform.dirty)
                m->transform.updateMatrices();
            if (m->bvhNodes.empty())
                throw std::runtime_error("Mesh BVH not built before TLAS");
            const DeviceBVHNode &root = m->bvhNodes[0];
            AABB local{vec3(root.bmin.x, root.bmin.y, root.bmin.z), vec3(root.bmax.x, root.bmax.y, root.bmax.z)};
            AABB world = m->transform.transformAABB(local);
            refs[i].id = (int)i;
            refs[i].b = world;
            refs[i].c = world.center();
        }
        h_tlasNodes.clear();
        h_tlasMeshIndices.clear();
        ptrt_detail::build_bvh_range(refs, 0, (int)refs.size(), bvhLeafTarget_ + bvhLeafTol_, h_tlasNodes,
                                     h_tlasMeshIndices);
    }

    // host half of updateAccelerationStructures (scene.cuh:596-743): rebuild dirty
    // BLAS, refresh descriptors, detect moved instances, rebuild the TLAS
    // `forRead`: somebody is about to read the host trees (flatten()).  A commit that stays on the GPU does not: the host copy
    // of a tree the GPU refitted / rebuilt is brought up to date only when it is read or uploaded again -- before a full upload,
    // for a TLAS with inner nodes (rebuilt on the host over the new boxes), when an instance moved (its world box comes from
    // the tree's root) -- not on every commit (a CPU refit of the whole tree, and after a GPU rebuild a read-back that waits
    // for the stream).
    void prepareHostStructures(bool forRead = true) {
        if (meshes.empty())
            return;
        bool need_trees = forRead || geometryDirty || flatMeshes.size() != meshes.size() || h_tlasNodes.size() != 1;
        for (size_t i = 0; i < meshes.size() && !need_trees; ++i) {
            const Mesh *m = meshes[i].get();
            need_trees = m->vertsDirty || m->bvhDirty || m->bvhNodes.empty() || m->transform.dirty ||
                         std::memcmp(&lastWorld[i], &m->transform.worldMatrix, sizeof(mat4)) != 0;
        }
        const bool synced = need_trees && syncHostTrees(); // (trees the GPU refitted / rebuilt since the host last looked)
        bool tlas_dirty = h_tlasNodes.empty() || synced;
        if (flatMeshes.size() != meshes.size()) {
            flatMeshes.assign(meshes.size(), ptrt_mesh_desc{});
            lastWorld.assign(meshes.size(), mat4());
            tlas_dirty = true;
            geometryDirty = materialsDirty = true;
        }
        for (size_t i = 0; i < meshes.size(); ++i) {
            Mesh *m = meshes[i].get();
            if (m->vertsDirty) {
                geometryDirty = true;
                m->vertsDirty = false;
            }
            if (m->bvhDirty || m->bvhNodes.empty()) {
                m->setBVHLeafParams(bvhLeafTarget_, bvhLeafTol_);
                m->buildBVH();
                tlas_dirty = true;
                geometryDirty = true;
            }
            bool moved = m->transform.dirty;
            if (moved)
                m->transform.updateMatrices();
            else
                moved = std::memcmp(&lastWorld[i], &m->transform.worldMatrix, sizeof(mat4)) != 0;
            if (moved) { // an instance moved: new matrices + TLAS, the triangles stay (ptrt_update_instances)
                tlas_dirty = true;
                instancesDirty = true;
            }
            lastWorld[i] = m->transform.worldMatrix;
            ptrt_mesh_desc &d = flatMeshes[i];
            d.verts = reinterpret_cast<const ptrt_vec3 *>(m->vertices.data());
            d.vert_count = (int)m->vertices.size();
            d.faces = m->faces.data();
            d.face_count = (int)m->faces.size();
            d.nodes = m->bvhNodes.data();
            d.node_count = (int)m->bvhNodes.size();
            d.prim_indices = m->bvhPrimIndices.data();
            d.prim_count = (int)m->bvhPrimIndices.size();
            std::memcpy(d.world, m->transform.worldMatrix.m, sizeof d.world);
            std::memcpy(d.inverse, m->transform.inverseMatrix.m, sizeof d.inverse);
            std::memcpy(d.normal, m->transform.normalMatrix.m, sizeof d.normal);
            d.has_transform = (m->transform.position.length() > 0.001f || m->transform.rotation.length() > 0.001f ||
                               fabsf(m->transform.scale.x - 1.0f) > 0.001f)
                                  ? 1
                                  : 0; // scene.cuh:718-721 (only scale.x is inspected)
        }
        if (tlas_dirty)
            buildTLAS();

        // materials: AoS -> SoA (scene.cuh:286-431)
        const size_t n = mesh_materials.size();
        auto c = [](const vec3 &a) { return ptrt_vec3{a.x, a.y, a.z}; };
        soa.albedo.resize(n); soa.specular.resize(n); soa.emission.resize(n); soa.subsurfaceColor.resize(n);
        soa.sheenTint.resize(n); soa.metallic.resize(n); soa.roughness.resize(n); soa.ior.resize(n);
        soa.transmission.resize(n); soa.transmissionRoughness.resize(n); soa.clearcoat.resize(n);
        soa.clearcoatRoughness.resize(n); soa.subsurfaceRadius.resize(n); soa.anisotropy.resize(n);
        soa.sheen.resize(n); soa.iridescence.resize(n); soa.iridescenceThickness.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const Material &m = mesh_materials[i];
            soa.albedo[i] = c(m.albedo); soa.specular[i] = c(m.specular); soa.emission[i] = c(m.emission);
            soa.subsurfaceColor[i] = c(m.subsurfaceColor); soa.sheenTint[i] = c(m.sheenTint);
            soa.metallic[i] = m.metallic; soa.roughness[i] = m.roughness; soa.ior[i] = m.ior;
            soa.transmission[i] = m.transmission; soa.transmissionRoughness[i] = m.transmissionRoughness;
            soa.clearcoat[i] = m.clearcoat; soa.clearcoatRoughness[i] = m.clearcoatRoughness;
            soa.subsurfaceRadius[i] = m.subsurfaceRadius; soa.anisotropy[i] = m.anisotropy; soa.sheen[i] = m.sheen;
            soa.iridescence[i] = m.iridescence; soa.iridescenceThickness[i] = m.iridescenceThickness;
        }
        flatLights.resize(lights.size());
        for (size_t i = 0; i < lights.size(); ++i) {
            const Light &l = lights[i];
            flatLights[i] = ptrt_light{(int32_t)l.type, c(l.position), c(l.direction), c(l.color), l.intensity,
                                       l.range, l.innerCone, l.outerCone, l.radius};
        }
        flat.meshes = flatMeshes.data();
        flat.mesh_count = (int)flatMeshes.size();
        flat.tlas_nodes = h_tlasNodes.data();
        flat.tlas_node_count = (int)h_tlasNodes.size();
        flat.tlas_mesh_indices = h_tlasMeshIndices.data();
        flat.tlas_index_count = (int)h_tlasMeshIndices.size();
        flat.materials = ptrt_materials{soa.albedo.data(), soa.specular.data(), soa.metallic.data(),
                                        soa.roughness.data(), soa.emission.data(), soa.ior.data(),
                                        soa.transmission.data(), soa.transmissionRoughness.data(),
                                        soa.clearcoat.data(), soa.clearcoatRoughness.data(),
                                        soa.subsurfaceColor.data(), soa.subsurfaceRadius.data(),
                                        soa.anisotropy.data(), soa.sheen.data(), soa.sheenTint.data(),
                                        soa.iridescence.data(), soa.iridescenceThickness.data(), (int32_t)n};
        flat.lights = flatLights.data();
        flat.light_count = (int)flatLights.size();
        flat.camera = camera.flat();
        flat.sky_top = c(sky_color_top);
        flat.sky_bottom = c(sky_color_bottom);
        flat.use_sky = use_sky ? 1 : 0;
        flat.env_rgba = envMap.empty() ? nullptr : envMap.data();
        flat.env_width = env_width;
        flat.env_height = env_height;
    }

    // The dynamic-geometry policy (setDynamicGeometryPolicy): meshes whose vertices changed while their topology is what
    // the device holds keep their uploaded tree.  Returns false when the commit has to take the reference's route.
    bool commitOnGpu() {
        if (dynPolicy == DynamicGeometryPolicy::HostRebuild || !gpu_resources_initialized || geometryDirty ||
            uploadedFaces.size() != meshes.size() || flatMeshes.size() != meshes.size())
            return false;
        std::vector<size_t> moved;
        for (size_t i = 0; i < meshes.size(); ++i) {
            const Mesh *m = meshes[i].get();
            if (!m->bvhDirty && !m->vertsDirty)
                continue;
            const std::vector<Tri> &uf = uploadedFaces[i];
            if (m->bvhNodes.empty() || m->vertices.size() != uploadedVerts[i] || m->faces.size() != uf.size())
                return false;
            const auto c0 = std::chrono::steady_clock::now();
            const bool same = sameFaces(*m, i);
            compareMicros += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count();
            if (!same)
                return false;
            moved.push_back(i);
        }
        if (moved.empty())
            return false;
        const bool rebuild = dynPolicy == DynamicGeometryPolicy::GpuRebuild;
        for (size_t i : moved) {
            Mesh *m = meshes[i].get();
            check(ptrt_update_vertices(ctx, (int)i, &m->vertices[0].x, (int)m->vertices.size(), 0), "Failed to update vertices");
            if (rebuild)
                check(ptrt_build_bvh(ctx, (int)i), "Failed to build BVH");
            m->bvhDirty = m->vertsDirty = false;
            // 1: boxes to refit on the host; 2: the prim order must be read back first (it stays pending across later refits)
            hostTreeStale[i] = rebuild ? 2 : (hostTreeStale[i] == 2 ? 2 : 1);
            flatMeshes[i].verts = reinterpret_cast<const ptrt_vec3 *>(m->vertices.data()); // (the caller may have re-allocated them)
            flatMeshes[i].faces = m->faces.data();
        }
        if (!rebuild)
            check(ptrt_refit(ctx), "Failed to refit");
        if (h_tlasNodes.size() > 1) { // a real TLAS is rebuilt on the host over the new boxes, as the reference's commit does
            syncHostTrees();
            syncTLAS();
        }
        ++gpuDynamicCommits;
        return true;
    }
    // Is the face list of mesh i what the device holds?  A `Triangles` mesh as updatePTScene rewrites it (PTRTtransfer.cuh:2249-2270)
    // is an unshared-vertex soup: face k = {3k, 3k+1, 3k+2}.  If the uploaded list was that soup (noted at upload), the new
    // list is checked against the PATTERN -- one pass over 12 bytes per face, no second array to stream -- else word by word
    // against the copy kept from the upload.
    bool sameFaces(const Mesh &m, size_t i) const {
        const std::vector<Tri> &uf = uploadedFaces[i];
        if (i < uploadedSoup.size() && uploadedSoup[i]) {
            const Tri *f = m.faces.data();
            const int n = (int)m.faces.size();
            int bad = 0;
            for (int k = 0; k < n; ++k) // (branch-free: the compiler vectorises it)
                bad |= (f[k].v0 ^ (3 * k)) | (f[k].v1 ^ (3 * k + 1)) | (f[k].v2 ^ (3 * k + 2));
            return bad == 0;
        }
        return std::memcmp(m.faces.data(), uf.data(), uf.size() * sizeof(Tri)) == 0;
    }
    // host copies of the trees the GPU refitted (boxes) or rebuilt (prim order, then boxes); true if any changed
    bool syncHostTrees() {
        bool any = false;
        for (size_t i = 0; i < hostTreeStale.size() && i < meshes.size(); ++i) {
            if (!hostTreeStale[i])
                continue;
            Mesh *m = meshes[i].get();
            if (hostTreeStale[i] == 2 && ctx)
                check(ptrt_read_prim_order(ctx, (int)i, m->bvhPrimIndices.data(), (int)m->bvhPrimIndices.size()),
                      "Failed to read the prim order");
            const bool bd = m->bvhDirty; // (a caller's later edit stays pending)
            m->refitBVH();
            m->bvhDirty = bd;
            hostTreeStale[i] = 0;
            any = true;
        }
        return any;
    }

    void updateAccelerationStructures() {
        if (meshes.empty())
            return;
        struct Timer {
            double &acc;
            std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
            ~Timer() { acc += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
        } timer{commitMicros};
        if (!ctx) { // host-only scene: the host half of a commit (trees, TLAS, material arrays); there is nothing to upload to
            prepareHostStructures();
            return;
        }
        commitOnGpu();
        prepareHostStructures(false);
        if (geometryDirty) {
            check(ptrt_upload_geometry(ctx, flat.meshes, flat.mesh_count, flat.tlas_nodes, flat.tlas_node_count,
                                       flat.tlas_mesh_indices, flat.tlas_index_count),
                  "Failed to upload geometry");
            geometryDirty = instancesDirty = false;
            ++geometryUploads;
            // what the device now holds, for the dynamic-geometry policy
            uploadedFaces.resize(meshes.size());
            uploadedVerts.resize(meshes.size());
            uploadedSoup.assign(meshes.size(), 0);
            hostTreeStale.assign(meshes.size(), 0);
            for (size_t i = 0; i < meshes.size(); ++i) {
                uploadedVerts[i] = meshes[i]->vertices.size();
                if (dynPolicy != DynamicGeometryPolicy::HostRebuild) {
                    uploadedFaces[i] = meshes[i]->faces;
                    const std::vector<Tri> &f = uploadedFaces[i];
                    bool soup = f.size() * 3 == meshes[i]->vertices.size();
                    for (size_t k = 0; soup && k < f.size(); ++k)
                        soup = f[k].v0 == (int)(3 * k) && f[k].v1 == (int)(3 * k + 1) && f[k].v2 == (int)(3 * k + 2);
                    uploadedSoup[i] = soup ? 1 : 0;
                } else {
                    uploadedFaces[i].clear(); // (the default policy pays nothing for the copies)
                }
            }
        } else if (instancesDirty) {
            check(ptrt_update_instances(ctx, flat.meshes, flat.mesh_count, flat.tlas_nodes, flat.tlas_node_count,
                                        flat.tlas_mesh_indices, flat.tlas_index_count),
                  "Failed to update instances");
            instancesDirty = false;
        }
        if (materialsDirty) {
            check(ptrt_upload_materials(ctx, &flat.materials), "Failed to upload materials");
            materialsDirty = false;
        }
        if (lightsDirty) {
            check(ptrt_upload_lights(ctx, flat.lights, flat.light_count), "Failed to upload lights");
            lightsDirty = false;
        }
        gpu_resources_initialized = true;
    }

    void renderInternal(unsigned char *pixels, int is_device) {
        // validateGPUResources: message on cerr and return without rendering (scene.cuh:216-251,1029-1031)
        if (meshes.empty()) {
            std::cerr << "ERROR: No meshes in scene!\n";
            return;
        }
        if (!gpu_resources_initialized) {
            std::cerr << "ERROR: Mesh descriptors not allocated!\n";
            return;
        }
        const bool denoise = perfSettings.enableDenoiser && denoiserAllocated;
        // bloom (scene.cuh:1137-1183) needs the whole frame and six non-empty mip levels
        const bool bloom = perfSettings.enableBloom && fullFrame && render_width >= 64 && render_height >= 64;
        if (((perfSettings.enableBloom && !bloom) || (!fullFrame && (perfSettings.enableDenoiser ||
                                                                     perfSettings.resolutionScale != 1.0f))) &&
            !warnedPost) {
            std::cerr << "NOTE: " << (fullFrame ? "bloom needs a frame of at least 64x64 pixels and is skipped\n"
                                                : "a band (tile) context renders its rows only: denoiser, bloom and "
                                                  "resolution scaling are skipped (apply them on the presenting rank)\n");
            warnedPost = true;
        }
        if (fullFrame)
            check(ptrt_set_bloom(ctx, bloom ? 1 : 0), "bloom");
        if (denoiserAllocated) { // scene.cuh:1103-1127: motion vectors + Denoiser::denoise after the trace
            check(ptrt_set_option(ctx, "denoiser_active", denoise ? 1 : 0), "denoiser option");
            check(ptrt_set_option(ctx, "motion_vectors", (denoise && perfSettings.enableMotionVectors) ? 1 : 0),
                  "denoiser option");
            check(ptrt_set_prev_view_proj(ctx, prev_view_proj.m), "prev view-proj");
        }
        updateAccelerationStructures();
        if (cameraDirty) {
            ptrt_camera c = camera.flat();
            check(ptrt_set_camera(ctx, &c), "Failed to set camera");
            cameraDirty = false;
        }
        if (skyDirty) {
            ptrt_vec3 t{sky_color_top.x, sky_color_top.y, sky_color_top.z},
                b{sky_color_bottom.x, sky_color_bottom.y, sky_color_bottom.z};
            check(ptrt_set_sky(ctx, &t, &b, use_sky ? 1 : 0), "Failed to set sky");
            skyDirty = false;
        }
        if (envDirty) {
            check(ptrt_set_env_map(ctx, envMap.empty() ? nullptr : envMap.data(), env_width, env_height),
                  "Failed to upload the environment map");
            envDirty = false;
        }
        int rc = ptrt_render(ctx, frame_count_, perfSettings.samplesPerPixel, perfSettings.maxBounceDepth, pixels,
                             is_device);
        if (rc != PTRT_OK) // launch errors are logged, not thrown (scene.cuh:1036-1042)
            std::cerr << "HIP kernel launch failed at path_trace_kernel: " << ptrt_last_error(ctx) << "\n";
        frame_count_++;
        prev_view_proj = camera.get_view_proj(); // scene.cuh:1208
    }
};

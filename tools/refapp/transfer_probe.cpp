// transfer_probe -- the reference's dynamic-geometry caller over the mirror.
//
// Compiles src/common/PTRTtransfer.cuh IN PLACE from /root/reference with -DUNIFIED_SCENE_ENABLE_PT -- UnifiedScene,
// UnifiedSceneBuilder::buildPTScene / updatePTScene / updatePTCamera -- against the MI355X mirror: the two project headers
// it includes are forwarded (tools/refapp/fwd: common/vec3.cuh -> ptrt/math.hpp, pathtracer/scene/scene.cuh ->
// ptrt/scene.hpp).  Nothing of the reference is copied, nothing it needs is stubbed.
//
// The scene: a `Triangles` mesh (a 12 x 12-cell sheet whose vertices move every step: the fluid-sim case), a dynamic cube (an
// instance whose transform moves), a static floor and a static sphere with a baked transform, two lights, a gradient
// sky.  Per step the application rewrites meshDesc.triangleVerts, moves the cube and the camera, and calls
//     updatePTScene(scene, unified);  updatePTCamera(scene, unified);
// -- updatePTScene rewrites mesh->vertices / faces, sets bvhDirty / vertsDirty (PTRTtransfer.cuh:2249-2270) and ends in
// scene.commitObjectChanges() (:2380, scene.cuh:1784).  That sequence is left exactly as the reference has it; the only
// line an application adds is scene->setDynamicGeometryPolicy(...) after buildPTScene.
//
//   transfer_probe golden <out.json>          build container, host-only scenes: the canonical byte stream (ptrt/serialize.hpp)
//                                             of the flattened scene after build and after each of two steps, default policy
//                                             (= the reference's host rebuild) -> tests/golden/transfer_scenes.json
//   transfer_probe gpu <policy 0|1|2> <out>   GPU box: the same sequence on device 0 with that policy; writes per frame the
//                                             scene's bytes, RGB8 and HDR frame, and the commit / upload counters, as one
//                                             binary file that tests/test_transfer_gpu.py compares with the oracle
static int g_probe_device = -1;
#define PTRT_DEFAULT_DEVICE g_probe_device // (Scene(w, h)'s device: buildPTScene constructs the scene itself)
#define UNIFIED_SCENE_ENABLE_PT
#include "common/PTRTtransfer.cuh"

#include "ptrt/serialize.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

constexpr int CELLS = 12, W = 320, H = 180, SPP = 2, DEPTH = 4, STEPS = 2;

// the sheet at time step k: plain fp32 products and sums in a fixed order (tests/test_transfer_*.py restate it in numpy)
std::vector<vec3> sheet(int k) {
    const float t = (float)k;
    std::vector<vec3> g((CELLS + 1) * (CELLS + 1));
    for (int j = 0; j <= CELLS; ++j)
        for (int i = 0; i <= CELLS; ++i) {
            const float x = -3.0f + 0.5f * (float)i, z = -3.0f + 0.5f * (float)j;
            const float a = x * z, b = x - z;
            const float y = (0.05f * a) + ((0.04f * t) * b) + (0.02f * t) * (x * x);
            g[j * (CELLS + 1) + i] = vec3(x, y, z);
        }
    std::vector<vec3> v;
    for (int j = 0; j < CELLS; ++j)
        for (int i = 0; i < CELLS; ++i) {
            const vec3 &A = g[j * (CELLS + 1) + i], &B = g[j * (CELLS + 1) + i + 1], &C = g[(j + 1) * (CELLS + 1) + i + 1],
                       &D = g[(j + 1) * (CELLS + 1) + i];
            for (const vec3 *p : {&A, &C, &B, &A, &D, &C}) // CCW from +Y
                v.push_back(*p);
        }
    return v;
}

UnifiedScene make_unified() {
    UnifiedScene u(W, H);
    u.setCamera(vec3(0.0f, 3.0f, 7.0f), vec3(0.0f, 0.0f, 0.0f), vec3(0.0f, 1.0f, 0.0f), 42.0f);
    u.setBVHParams(4, 2);
    UnifiedMeshDesc water;
    water.type = UnifiedMeshDesc::Type::Triangles;
    water.triangleVerts = sheet(0);
    water.material = UnifiedMaterial(vec3(0.2f, 0.45f, 0.8f), 0.15f, 0.0f);
    water.name = "water";
    u.addMesh(water);
    UnifiedMeshDesc cube = UnifiedMeshDesc::Cube(UnifiedMaterial::Gold());
    cube.setPosition(vec3(-1.5f, 0.9f, 0.5f)).setRotation(vec3(0.2f, 0.4f, 0.0f)).setScale(0.8f).setDynamic(true);
    cube.name = "cube";
    u.addMesh(cube);
    u.addMesh(UnifiedMeshDesc::PlaneXZ(-1.0f, 6.0f, UnifiedMaterial(vec3(0.7f, 0.7f, 0.65f), 0.9f, 0.0f)));
    UnifiedMeshDesc ball = UnifiedMeshDesc::Sphere(12, UnifiedMaterial(vec3(0.8f, 0.3f, 0.25f), 0.4f, 0.0f));
    ball.setScale(vec3(1.2f, 0.9f, 1.2f)).setRotation(vec3(0.0f, 0.3f, 0.1f)).setPosition(vec3(1.6f, 0.8f, -0.5f));
    u.addMesh(ball);
    u.addPointLight(vec3(0.0f, 5.0f, 2.0f), vec3(1.0f, 0.95f, 0.9f), 40.0f, 100.0f, 0.3f);
    u.addSpotLight(vec3(-3.0f, 4.0f, 3.0f), vec3(0.6f, -0.8f, -0.6f), vec3(0.6f, 0.7f, 1.0f), 60.0f, 0.3f, 0.5f, 50.0f, 0.0f);
    u.setSkyGradient(vec3(0.3f, 0.5f, 0.9f), vec3(0.9f, 0.9f, 1.0f));
    return u;
}

// what the application does between frames
void step(UnifiedScene &u, int k) {
    u.meshes[0].triangleVerts = sheet(k);
    u.markMeshDirty(0);
    u.meshes[1].transform.setPosition(vec3(-1.5f + 0.4f * (float)k, 0.9f, 0.5f));
    u.meshes[1].transform.setRotation(vec3(0.2f, 0.4f + 0.3f * (float)k, 0.0f));
    u.markMeshDirty(1);
    u.setCamera(vec3(0.5f * (float)k, 3.0f, 7.0f), vec3(0.0f, 0.0f, 0.0f), vec3(0.0f, 1.0f, 0.0f), 42.0f);
}

std::vector<uint8_t> bytes_of(Scene &scene) { return ptrt_detail::serialize_scene(scene.flatten()); }

void put_blob(FILE *f, const void *p, uint64_t n) {
    std::fwrite(&n, 8, 1, f);
    std::fwrite(p, 1, (size_t)n, f);
}

int golden(const char *path) {
    FILE *out = std::fopen(path, "w");
    if (!out)
        return 3;
    UnifiedScene u = make_unified();
    std::unique_ptr<Scene> scene = UnifiedSceneBuilder::buildPTScene(u);
    std::fprintf(out, "{\n \"width\": %d, \"height\": %d, \"cells\": %d, \"steps\": [\n", W, H, CELLS);
    for (int k = 0; k <= STEPS; ++k) {
        if (k > 0) {
            step(u, k);
            UnifiedSceneBuilder::updatePTScene(*scene, u); // ends in scene.commitObjectChanges(): host half only on a host-only scene
            UnifiedSceneBuilder::updatePTCamera(*scene, u);
        }
        const std::vector<uint8_t> b = bytes_of(*scene);
        std::fprintf(out, "  {\"step\": %d, \"meshes\": %zu, \"bytes\": %zu, \"hex\": \"", k, scene->getMeshCount(), b.size());
        for (uint8_t c : b)
            std::fprintf(out, "%02x", c);
        std::fprintf(out, "\"}%s\n", k == STEPS ? "" : ",");
    }
    std::fprintf(out, " ]\n}\n");
    std::fclose(out);
    return 0;
}

int gpu(int policy, const char *path) {
    g_probe_device = 0;
    FILE *out = std::fopen(path, "wb");
    if (!out)
        return 3;
    UnifiedScene u = make_unified();
    std::unique_ptr<Scene> scene = UnifiedSceneBuilder::buildPTScene(u);
    scene->setDynamicGeometryPolicy((Scene::DynamicGeometryPolicy)policy); // the ONE added line
    scene->setPerfSamplesPerPixel(SPP);
    scene->setMaxBounceDepth(DEPTH);
    scene->setDenoiserEnabled(false);
    scene->setBloomEnabled(false);
    scene->initBlueNoise();
    scene->uploadToGPU();
    const uint32_t header[6] = {0x54525046u /* "FPRT" */, (uint32_t)W, (uint32_t)H, (uint32_t)(STEPS + 1), (uint32_t)SPP, (uint32_t)DEPTH};
    std::fwrite(header, 4, 6, out);
    std::vector<unsigned char> rgb((size_t)W * H * 3);
    std::vector<float> accum((size_t)W * H * 3);
    for (int k = 0; k <= STEPS; ++k) {
        if (k > 0) {
            step(u, k);
            UnifiedSceneBuilder::updatePTScene(*scene, u);
            UnifiedSceneBuilder::updatePTCamera(*scene, u);
        }
        scene->render_to_host(rgb.data());
        if (ptrt_read_buffer(scene->backend(), PTRT_BUF_ACCUM, accum.data(), accum.size() * sizeof(float)) != PTRT_OK)
            return 4;
        const uint64_t counts[2] = {scene->gpuDynamicCommitCount(), scene->geometryUploadCount()};
        std::fwrite(counts, 8, 2, out);
        const std::vector<uint8_t> b = bytes_of(*scene); // (after the frame: brings a tree the GPU refitted up to date on the host)
        put_blob(out, b.data(), b.size());
        put_blob(out, rgb.data(), rgb.size());
        put_blob(out, accum.data(), accum.size() * sizeof(float));
    }
    std::fclose(out);
    std::printf("transfer_probe: policy %d, %d frames, %zu GPU commits, %zu geometry uploads\n", policy, STEPS + 1,
                scene->gpuDynamicCommitCount(), scene->geometryUploadCount());
    return 0;
}

} // namespace

int main(int argc, char **argv) {
    try {
        if (argc == 3 && !std::strcmp(argv[1], "golden"))
            return golden(argv[2]);
        if (argc == 4 && !std::strcmp(argv[1], "gpu"))
            return gpu(std::atoi(argv[2]), argv[3]);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "transfer_probe: %s\n", e.what());
        return 5;
    }
    std::fprintf(stderr, "usage: transfer_probe golden <out.json> | gpu <policy 0|1|2> <out.bin>\n");
    return 2;
}

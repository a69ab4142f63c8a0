// ptrt/serialize.hpp -- canonical byte stream of a flattened scene (ptrt_scene_desc).
//
// Two scenes have the same stream iff the back end receives the same arrays: per mesh vertices, faces, BLAS
// nodes, prim order, the three matrices and has_transform; TLAS; the 17 material arrays; lights; camera; sky.
// Used to pin scene builders against each other -- the reference's own buildSceneById compiled over this
// mirror (tools/refapp) against the Python recipes the tests and bench use (tests/test_refapp_scenes.py).
#pragma once
#include "../../../include/ptrt.h"

#include <cstdint>
#include <cstring>
#include <vector>

namespace ptrt_detail {
inline void put(std::vector<uint8_t> &o, const void *p, size_t n) {
    const uint8_t *b = static_cast<const uint8_t *>(p);
    o.insert(o.end(), b, b + n);
}
inline void put_i32(std::vector<uint8_t> &o, int32_t v) { put(o, &v, 4); }

inline std::vector<uint8_t> serialize_scene(const ptrt_scene_desc &d) {
    std::vector<uint8_t> o;
    put_i32(o, d.mesh_count);
    for (int m = 0; m < d.mesh_count; ++m) {
        const ptrt_mesh_desc &M = d.meshes[m];
        put_i32(o, M.vert_count);
        put(o, M.verts, (size_t)M.vert_count * sizeof(ptrt_vec3));
        put_i32(o, M.face_count);
        put(o, M.faces, (size_t)M.face_count * sizeof(ptrt_tri));
        put_i32(o, M.node_count);
        put(o, M.nodes, (size_t)M.node_count * sizeof(ptrt_bvh_node));
        put_i32(o, M.prim_count);
        put(o, M.prim_indices, (size_t)M.prim_count * 4);
        put(o, M.world, sizeof M.world);
        put(o, M.inverse, sizeof M.inverse);
        put(o, M.normal, sizeof M.normal);
        put_i32(o, M.has_transform);
    }
    put_i32(o, d.tlas_node_count);
    put(o, d.tlas_nodes, (size_t)d.tlas_node_count * sizeof(ptrt_bvh_node));
    put_i32(o, d.tlas_index_count);
    put(o, d.tlas_mesh_indices, (size_t)d.tlas_index_count * 4);
    const ptrt_materials &t = d.materials;
    const size_t n = (size_t)t.count;
    put_i32(o, t.count);
    for (const ptrt_vec3 *v : {t.albedo, t.specular, t.emission, t.subsurface_color, t.sheen_tint})
        put(o, v, n * sizeof(ptrt_vec3));
    for (const float *f : {t.metallic, t.roughness, t.ior, t.transmission, t.transmission_roughness, t.clearcoat,
                           t.clearcoat_roughness, t.subsurface_radius, t.anisotropy, t.sheen, t.iridescence,
                           t.iridescence_thickness})
        put(o, f, n * 4);
    put_i32(o, d.light_count);
    put(o, d.lights, (size_t)d.light_count * sizeof(ptrt_light));
    put(o, &d.camera, sizeof d.camera);
    put(o, &d.sky_top, sizeof d.sky_top);
    put(o, &d.sky_bottom, sizeof d.sky_bottom);
    put_i32(o, d.use_sky);
    put_i32(o, d.env_width);
    put_i32(o, d.env_height);
    if (d.env_rgba)
        put(o, d.env_rgba, (size_t)d.env_width * d.env_height * 16);
    return o;
}
} // namespace ptrt_detail

// ptrt/mesh.hpp -- host-side triangle mesh + BLAS builder of the Scene API.
//
// Mirrors `class Mesh` (reference: src/pathtracer/scene/mesh.cuh:49-232) minus the
// device pointers: all device memory belongs to the back-end context
// (include/ptrt.h), the mesh only keeps host arrays and dirty flags.
#pragma once
#include "../../../include/ptrt.h"
#include "math.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

// src/common/triangle.cuh:15-92 -- the input type of Scene::addTriangles (which reads v0, v1, v2).  The other public
// members are kept for callers that use them: pre-computed edges, the unnormalised normal, and the two-sided
// Moeller-Trumbore test of the OLD intersection code (host arithmetic; the render path has its own, intersection.cuh:219).
struct Triangle {
    vec3 v0, v1, v2;
    vec3 e1, e2; // v1 - v0, v2 - v0
    vec3 n;      // cross(e1, e2), not unit length
    Triangle() = default;
    Triangle(const vec3 &a, const vec3 &b, const vec3 &c) : v0(a), v1(b), v2(c) {
        e1 = v1 - v0;
        e2 = v2 - v0;
        n = cross(e1, e2);
    }
    vec3 normal() const { return normalize(n); }
    float area() const { return 0.5f * length(n); }
    void bounds(vec3 &bmin, vec3 &bmax) const {
        bmin.x = fminf(v0.x, fminf(v1.x, v2.x));
        bmin.y = fminf(v0.y, fminf(v1.y, v2.y));
        bmin.z = fminf(v0.z, fminf(v1.z, v2.z));
        bmax.x = fmaxf(v0.x, fmaxf(v1.x, v2.x));
        bmax.y = fmaxf(v0.y, fmaxf(v1.y, v2.y));
        bmax.z = fmaxf(v0.z, fmaxf(v1.z, v2.z));
    }
    bool intersect(const Ray &ray, float &t, float &u, float &v) const { // two-sided (RT_CULL_BACKFACES 0, the default)
        const float EPS = 1e-6f;
        const vec3 pvec = cross(ray.direction(), e2);
        const float det = dot(e1, pvec);
        if (fabsf(det) < EPS)
            return false;
        const float invDet = 1.0f / det;
        const vec3 tvec = ray.origin() - v0;
        u = dot(tvec, pvec) * invDet;
        if (u < 0.0f || u > 1.0f)
            return false;
        const vec3 qvec = cross(tvec, e1);
        v = dot(ray.direction(), qvec) * invDet;
        if (v < 0.0f || (u + v) > 1.0f)
            return false;
        t = dot(e2, qvec) * invDet;
        return t > EPS;
    }
};

using Tri = ptrt_tri;                // mesh.cuh:45-47
using DeviceBVHNode = ptrt_bvh_node; // mesh.cuh:37-43 (same 40-byte layout)

namespace ptrt_detail {
struct BuildRef {
    int id;
    vec3 c;
    AABB b;
};
// Median split on the longest CENTROID axis, leaf when n <= leafMax, nodes in
// pre-order, leaf primitives appended in range order (mesh.cuh:403-492 for
// triangles, scene.cuh:458-594 for the TLAS).  std::nth_element makes the
// topology libstdc++-specific; it is an input of the path, not part of it.
inline int build_bvh_range(std::vector<BuildRef> &R, int begin, int end, int leafMax,
                           std::vector<DeviceBVHNode> &nodes, std::vector<int> &prims) {
    AABB bb = AABB::make_invalid(), cb = AABB::make_invalid();
    for (int i = begin; i < end; ++i) {
        bb.expand(R[i].b);
        cb.expand(R[i].c);
    }
    const int n = end - begin;
    const int me = (int)nodes.size();
    nodes.emplace_back();
    nodes[me].bmin = {bb.bmin.x, bb.bmin.y, bb.bmin.z};
    nodes[me].bmax = {bb.bmax.x, bb.bmax.y, bb.bmax.z};
    nodes[me].left = nodes[me].right = nodes[me].start = -1;
    nodes[me].count = 0;
    if (n <= leafMax) {
        nodes[me].start = (int)prims.size();
        nodes[me].count = n;
        for (int i = begin; i < end; ++i)
            prims.push_back(R[i].id);
        return me;
    }
    vec3 e = cb.extent();
    const int axis = (e.x > e.y && e.x > e.z) ? 0 : ((e.y > e.z) ? 1 : 2);
    const int mid = (begin + end) / 2;
    std::nth_element(R.begin() + begin, R.begin() + mid, R.begin() + end,
                     [axis](const BuildRef &A, const BuildRef &B) { return A.c[axis] < B.c[axis]; });
    const int L = build_bvh_range(R, begin, mid, leafMax, nodes, prims);
    const int Rn = build_bvh_range(R, mid, end, leafMax, nodes, prims);
    nodes[me].left = L;
    nodes[me].right = Rn;
    return me;
}
} // namespace ptrt_detail

class Mesh {
  public:
    std::vector<vec3> vertices;
    std::vector<Tri> faces;
    std::vector<DeviceBVHNode> bvhNodes;
    std::vector<int> bvhPrimIndices;
    bool bvhDirty = true;
    bool vertsDirty = true;
    int bvhLeafTarget = 12; // mesh.cuh:65-66
    int bvhLeafTol = 5;
    Transform3D transform;
    AABB localAABB = AABB::make_invalid();

    // unit cube centred at the origin, 8 vertices / 12 faces (mesh.cuh:221-229)
    Mesh() {
        for (int i = 0; i < 8; ++i) {
            const bool xh = ((i & 3) == 1) || ((i & 3) == 2);
            vertices.emplace_back(xh ? 0.5f : -0.5f, (i & 2) ? 0.5f : -0.5f, (i & 4) ? 0.5f : -0.5f);
        }
        static const int F[12][3] = {{0, 2, 1}, {0, 3, 2}, {4, 5, 6}, {4, 6, 7}, {0, 1, 5}, {0, 5, 4},
                                     {3, 7, 6}, {3, 6, 2}, {0, 4, 7}, {0, 7, 3}, {1, 2, 6}, {1, 6, 5}};
        for (auto &f : F)
            faces.push_back({f[0], f[1], f[2]});
    }

    // Wavefront OBJ (the reader behind Scene::addMesh, mesh.cuh:238-323).  What the reference accepts is kept -- only
    // `v x y z` and `f ...` records count; a face corner is `i`, `i/t`, `i//n` or `i/t/n` of which only `i` is used,
    // 1-based or negative = relative to the vertices read so far; polygons are fanned around their first corner;
    // the vertices are re-centred on their mean (summed in double); the three error messages -- but the file is read
    // in one piece and scanned with a cursor instead of line by line through string streams.
    explicit Mesh(const std::string &path) {
        std::string text;
        {
            std::ifstream in(path, std::ios::binary);
            if (!in)
                throw std::runtime_error("Mesh: cannot open " + path);
            in.seekg(0, std::ios::end);
            const std::streamoff size = in.tellg();
            in.seekg(0, std::ios::beg);
            text.resize(size > 0 ? (size_t)size : 0);
            if (!text.empty())
                in.read(&text[0], (std::streamsize)text.size());
            if (in.bad())
                throw std::runtime_error("Mesh: hardware error reading " + path);
            text.resize((size_t)in.gcount());
        }
        struct Cursor {
            const char *p, *eol;
            void blanks() {
                while (p < eol && (*p == ' ' || *p == '\t' || *p == '\r'))
                    ++p;
            }
            // a decimal real as `istream >> float` takes it: [sign] digits [. digits] [e|E [sign] digits] -- the token is
            // delimited HERE, so strtof never sees the "nan", "inf" / "infinity" or hexadecimal forms it would accept and
            // the stream extraction of the reference refuses ("0x1p3" reads as 0 and the record then fails at the 'x');
            // an overflowing value fails like the extraction's failbit does
            bool real(float &out) {
                blanks();
                const char *q = p;
                if (q < eol && (*q == '-' || *q == '+'))
                    ++q;
                const char *d0 = q;
                while (q < eol && *q >= '0' && *q <= '9')
                    ++q;
                int mant = (int)(q - d0);
                if (q < eol && *q == '.') {
                    const char *f0 = ++q;
                    while (q < eol && *q >= '0' && *q <= '9')
                        ++q;
                    mant += (int)(q - f0);
                }
                if (mant == 0)
                    return false;
                if (q < eol && (*q == 'e' || *q == 'E')) {
                    const char *e = q + 1;
                    if (e < eol && (*e == '-' || *e == '+'))
                        ++e;
                    if (e < eol && *e >= '0' && *e <= '9') {
                        while (e < eol && *e >= '0' && *e <= '9')
                            ++e;
                        q = e;
                    }
                }
                char buf[64];
                const size_t n = (size_t)(q - p);
                float v;
                if (n < sizeof buf) {
                    std::memcpy(buf, p, n);
                    buf[n] = 0;
                    v = std::strtof(buf, nullptr);
                } else {
                    v = std::strtof(std::string(p, n).c_str(), nullptr);
                }
                if (!(v - v == 0.0f)) // overflow to infinity
                    return false;
                p = q;
                out = v;
                return true;
            }
            bool integer(int &out) {
                blanks();
                const char *q = p;
                bool neg = false;
                if (q < eol && (*q == '-' || *q == '+'))
                    neg = *q++ == '-';
                if (q >= eol || *q < '0' || *q > '9')
                    return false;
                long long v = 0;
                bool over = false;
                while (q < eol && *q >= '0' && *q <= '9') {
                    v = v * 10 + (*q++ - '0');
                    over = over || v > 2147483648ll; // (from here on the value no longer matters: no overflow of v itself)
                    if (over)
                        v = 0;
                }
                p = q;
                if (over || (neg ? -v : v) > 2147483647ll)
                    return false; // `istream >> int` sets failbit on a value outside int: the face record ends here
                out = (int)(neg ? -v : v);
                return true;
            }
        };
        double sum[3] = {0.0, 0.0, 0.0};
        std::vector<int> corner;
        const char *const begin = text.c_str(), *const end = begin + text.size();
        for (const char *line = begin; line < end;) {
            const char *eol = static_cast<const char *>(std::memchr(line, '\n', (size_t)(end - line)));
            if (!eol)
                eol = end;
            Cursor c{line, eol};
            line = eol + 1;
            c.blanks();
            const char *key = c.p;
            while (c.p < c.eol && *c.p != ' ' && *c.p != '\t' && *c.p != '\r')
                ++c.p;
            if (c.p - key != 1)
                continue; // comments, vn / vt / usemtl / ... and empty lines
            if (*key == 'v') {
                float x, y, z;
                if (c.real(x) && c.real(y) && c.real(z)) {
                    vertices.emplace_back(x, y, z);
                    sum[0] += x;
                    sum[1] += y;
                    sum[2] += z;
                }
            } else if (*key == 'f') {
                corner.clear();
                int id;
                while (c.integer(id)) {
                    corner.push_back(id < 0 ? (int)vertices.size() + id : id - 1);
                    while (c.p < c.eol && *c.p == '/') { // texture / normal references: skipped
                        ++c.p;
                        if (c.p < c.eol && *c.p == '/')
                            ++c.p;
                        int unused;
                        const char *at = c.p;
                        if (!(c.p < c.eol && *c.p != ' ' && *c.p != '\t' && c.integer(unused)))
                            c.p = at;
                    }
                }
                for (size_t k = 2; k < corner.size(); ++k)
                    faces.push_back({corner[0], corner[k - 1], corner[k]});
            }
        }
        if (vertices.empty() || faces.empty())
            throw std::runtime_error("Mesh: no valid geometry in " + path);
        const double n = (double)vertices.size();
        const float cx = (float)(sum[0] / n), cy = (float)(sum[1] / n), cz = (float)(sum[2] / n);
        for (vec3 &v : vertices) {
            v.x -= cx;
            v.y -= cy;
            v.z -= cz;
        }
    }

    Mesh(const Mesh &) = delete;
    Mesh &operator=(const Mesh &) = delete;

    size_t faceCount() const { return faces.size(); }
    size_t vertexCount() const { return vertices.size(); }

    void setBVHLeafParams(int target, int tol = 5) {
        bvhLeafTarget = target < 1 ? 1 : target;
        bvhLeafTol = tol < 0 ? 0 : tol;
        bvhDirty = true;
    }

    // mesh.cuh:403-492
    void buildBVH() {
        bvhNodes.clear();
        bvhPrimIndices.clear();
        if (faces.empty()) {
            bvhDirty = false;
            return;
        }
        std::vector<ptrt_detail::BuildRef> refs;
        refs.reserve(faces.size());
        for (int i = 0; i < (int)faces.size(); ++i) {
            const vec3 &a = vertices[faces[i].v0], &b = vertices[faces[i].v1], &c = vertices[faces[i].v2];
            ptrt_detail::BuildRef r;
            r.id = i;
            r.b = {a, a};
            r.b.expand(b);
            r.b.expand(c);
            r.c = (a + b + c) * (1.0f / 3.0f);
            refs.push_back(r);
        }
        ptrt_detail::build_bvh_range(refs, 0, (int)refs.size(), bvhLeafTarget + bvhLeafTol, bvhNodes,
                                     bvhPrimIndices);
        bvhDirty = false;
    }

    // Refit (NOT in the reference, which rebuilds every dirty mesh: mesh.cuh:403-492): keep the
    // tree and the leaf assignment, recompute every box from the current vertices.  Nodes are in
    // pre-order (children after parents), so one reverse sweep is bottom-up.  The GPU refit
    // (csrc/pt_refit.hip.h) produces the same boxes bit for bit: min/max are exact.
    void refitBVH() {
        for (int i = (int)bvhNodes.size() - 1; i >= 0; --i) {
            DeviceBVHNode &n = bvhNodes[i];
            AABB b = AABB::make_invalid();
            if (n.count > 0) {
                for (int k = 0; k < n.count; ++k) {
                    const Tri &t = faces[bvhPrimIndices[n.start + k]];
                    b.expand(vertices[t.v0]);
                    b.expand(vertices[t.v1]);
                    b.expand(vertices[t.v2]);
                }
            } else {
                for (int ch : {n.left, n.right})
                    if (ch >= 0) {
                        const DeviceBVHNode &c = bvhNodes[ch];
                        b.expand(AABB{vec3(c.bmin.x, c.bmin.y, c.bmin.z), vec3(c.bmax.x, c.bmax.y, c.bmax.z)});
                    }
            }
            n.bmin = {b.bmin.x, b.bmin.y, b.bmin.z};
            n.bmax = {b.bmax.x, b.bmax.y, b.bmax.z};
        }
        bvhDirty = false;
    }

    AABB boundingBox() const { // mesh.cuh:526-541
        if (vertices.empty())
            return {vec3(0.0f), vec3(0.0f)};
        AABB b{vertices[0], vertices[0]};
        for (size_t i = 1; i < vertices.size(); ++i)
            b.expand(vertices[i]);
        return b;
    }

    // vertex-baking transforms (mesh.cuh:548-640); each marks BVH + vertices dirty
    void scale(float s) { scale(vec3(s)); }
    void scale(vec3 s) {
        for (auto &v : vertices) {
            v.x *= s.x; v.y *= s.y; v.z *= s.z;
        }
        touch();
    }
    void translate(const vec3 &d) {
        for (auto &v : vertices)
            v = v + d;
        touch();
    }
    void moveTo(const vec3 &p) {
        AABB bb = boundingBox();
        translate(p - (bb.bmin + bb.bmax) * 0.5f);
    }
    void rotateSelfEulerXYZ(const vec3 &rad) {
        AABB bb = boundingBox();
        const vec3 c = (bb.bmin + bb.bmax) * 0.5f;
        const float cx = cosf(rad.x), sx = sinf(rad.x), cy = cosf(rad.y), sy = sinf(rad.y), cz = cosf(rad.z),
                    sz = sinf(rad.z);
        for (auto &v : vertices) {
            vec3 p = v - c;
            const float y1 = cx * p.y - sx * p.z, z1 = sx * p.y + cx * p.z; // about X
            p.y = y1; p.z = z1;
            const float x2 = cy * p.x + sy * p.z, z2 = -sy * p.x + cy * p.z; // about Y
            p.x = x2; p.z = z2;
            const float x3 = cz * p.x - sz * p.y, y3 = sz * p.x + cz * p.y; // about Z
            p.x = x3; p.y = y3;
            v = p + c;
        }
        touch();
    }

    // instance transform (mesh.cuh:165-197)
    void setTransform(const Transform3D &t) { transform = t; transform.dirty = true; transform.updateMatrices(); }
    void setPosition(const vec3 &p) { transform.setPosition(p); transform.updateMatrices(); }
    void setRotation(const vec3 &r) { transform.setRotation(r); transform.updateMatrices(); }
    void computeLocalAABB() { localAABB = boundingBox(); }
    AABB getWorldAABB() const {
        if (!transform.dirty && localAABB.bmin.x < 1e20f)
            return transform.transformAABB(localAABB);
        return boundingBox();
    }

  private:
    void touch() { bvhDirty = true; vertsDirty = true; }
};

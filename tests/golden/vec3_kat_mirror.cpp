// Test helper (built by tests/test_ref_probe.py with g++): the vec3 known-answer rows of oracle/ref_probe.cpp
// evaluated with the host mirror's vec3 (ptrt-game-engine_amd/host/ptrt/math.hpp) -- same seeded inputs, same
// expressions, results as bit patterns, one row per line.
#include "ptrt/math.hpp"

#include <cstdint>
#include <cstdio>
#include <cstring>


static uint32_t bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

int main() {
    uint32_t st = 12345u;
    auto rnd = [&]() {
        st = st * 1664525u + 1013904223u;
        return ((float)(st >> 8) / 16777216.0f - 0.5f) * 8.0f;
    };
    for (int k = 0; k < 48; ++k) {
        const float ax = rnd(), ay = rnd(), az = rnd(), bx = rnd(), by = rnd(), bz = rnd(), t = rnd();
        const vec3 a(ax, ay, az), b(bx, by, bz);
        const vec3 c = cross(a, b), n = b.normalized(), r = normalize(a - b), l = lerp(a, b, 0.5f + 0.1f * t), q = a * b + t * a - b / 3.0f;
        printf("%u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u\n", bits(dot(a, b)), bits(a.length()),
               bits(a.length_squared()), bits(c.x), bits(c.y), bits(c.z), bits(n.x), bits(n.y), bits(n.z), bits(r.x), bits(r.y),
               bits(r.z), bits(l.x), bits(l.y), bits(l.z), bits(q.x), bits(q.y), bits(q.z));
    }
    return 0;
}

// view.hpp -- the rtgl:: presentation calls of the reference's viewer
// (src/common/glfw_view_interop.hpp:43-374) over the HIP presentation ring (ptrt_present_*).
//
// The reference's frame loop is
//     uint8_t *d = rtgl::map_pbo_device_ptr(V);   // CUDA maps the GL pixel-buffer object
//     scene.render_to_device(d);
//     rtgl::unmap_pbo(V);
//     rtgl::blit_pbo_to_texture(V);               // glTexSubImage2D from the bound PBO
//     rtgl::draw_interop(V);                      // textured quad, swap, poll
// MI355X has no GL interop.  The same five calls here drive a ring of device frames mirrored into
// pinned host memory: map = next ring slot's device pointer, unmap = asynchronous device->host
// copy on the scene's stream, blit = wait for the OLDEST frame in flight and expose its host
// pixels (V.host_frame, RGB8 bottom-up exactly as the PBO held them -- the pointer a GL build
// hands to glTexSubImage2D(..., GL_RGB, GL_UNSIGNED_BYTE, V.host_frame)), draw = present it.
// With the default two slots the copy of frame i overlaps the rendering of frame i+1; the
// picture shown lags the render by slots-1 frames.
// This header is the HEADLESS build (no window system in the target image): draw_interop counts
// frames and can dump every n-th one as a binary PPM.
#pragma once
#include "scene.hpp"

#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>

namespace rtgl {

struct InteropViewer {
    ptrt_ctx *ctx = nullptr;
    int viewW = 0, viewH = 0;
    int slots = 2;
    int mapped = -1;                // slot handed out by map_pbo_device_ptr
    unsigned long long submitted = 0, presented = 0;
    const unsigned char *host_frame = nullptr; // set by blit_pbo_to_texture (NULL while the ring fills)
    std::string dump_prefix;        // "" = no files; else <prefix>NNNNNN.ppm
    int dump_every = 0;
};

inline void check(InteropViewer &V, int rc, const char *what) {
    if (rc != PTRT_OK)
        throw std::runtime_error(std::string(what) + ": " + ptrt_last_error(V.ctx));
}

// init_interop_viewer(V, width, height, title, cudaDevice) of the reference; the device is the scene's
inline void init_interop_viewer(InteropViewer &V, Scene &scene, const char * /*title*/ = "", int slots = 2) {
    V.ctx = scene.backend();
    if (!V.ctx)
        throw std::runtime_error("init_interop_viewer: host-only Scene has no GPU back end");
    V.viewW = scene.getWidth();
    V.viewH = scene.getHeight();
    V.slots = slots;
    V.mapped = -1;
    V.submitted = V.presented = 0;
    V.host_frame = nullptr;
    check(V, ptrt_present_create(V.ctx, slots), "ptrt_present_create failed");
}

inline uint8_t *map_pbo_device_ptr(InteropViewer &V, size_t *nbytes = nullptr) {
    void *p = nullptr;
    V.mapped = (int)(V.submitted % (unsigned long long)V.slots);
    check(V, ptrt_present_map(V.ctx, V.mapped, &p), "ptrt_present_map failed");
    if (nbytes)
        *nbytes = (size_t)V.viewW * V.viewH * 3;
    return static_cast<uint8_t *>(p);
}

inline void unmap_pbo(InteropViewer &V) {
    if (V.mapped < 0)
        throw std::runtime_error("unmap_pbo: nothing mapped");
    check(V, ptrt_present_unmap(V.ctx, V.mapped), "ptrt_present_unmap failed");
    V.mapped = -1;
    V.submitted++;
}

// the oldest frame in flight becomes V.host_frame once the ring is full (or `flush` is set)
inline void blit_pbo_to_texture(InteropViewer &V, bool flush = false) {
    V.host_frame = nullptr;
    const unsigned long long lag = flush ? 1ull : (unsigned long long)V.slots;
    if (V.submitted - V.presented < lag)
        return;
    const int slot = (int)(V.presented % (unsigned long long)V.slots);
    check(V, ptrt_present_acquire(V.ctx, slot, &V.host_frame), "ptrt_present_acquire failed");
    V.presented++;
}

inline void draw_interop(const InteropViewer &V) {
    if (!V.host_frame || V.dump_every <= 0 || V.dump_prefix.empty() || (V.presented - 1) % (unsigned long long)V.dump_every)
        return;
    char name[32];
    std::snprintf(name, sizeof name, "%06llu.ppm", V.presented - 1);
    std::ofstream f(V.dump_prefix + name, std::ios::binary);
    if (!f)
        throw std::runtime_error("draw_interop: cannot write " + V.dump_prefix + name);
    f << "P6\n" << V.viewW << ' ' << V.viewH << "\n255\n";
    for (int y = V.viewH - 1; y >= 0; --y) // the frame is bottom-up (scene.cuh:2014), a PPM top-down
        f.write(reinterpret_cast<const char *>(V.host_frame) + (size_t)y * V.viewW * 3, (std::streamsize)V.viewW * 3);
}

inline void destroy_interop_viewer(InteropViewer &V) {
    if (V.ctx)
        (void)ptrt_present_destroy(V.ctx);
    V = InteropViewer{};
}

} // namespace rtgl

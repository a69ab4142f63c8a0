// view.hpp -- the rtgl:: presentation calls of the reference's viewer
// (src/common/glfw_view_interop.hpp:43-374) over a HIP presentation ring (ptrt_ring_*).
//
// The reference's frame loop is
//     rtgl::init_interop_viewer(V, width, height, title, cudaDevice);
//     uint8_t *d = rtgl::map_pbo_device_ptr(V);   // CUDA maps the GL pixel-buffer object
//     scene.render_to_device(d);
//     rtgl::unmap_pbo(V);
//     rtgl::blit_pbo_to_texture(V);               // glTexSubImage2D from the bound PBO
//     rtgl::draw_interop(V);                      // textured quad, swap, poll
// and compiles unchanged against this header.  MI355X has no GL interop, so the PBO is a ring of
// device frames mirrored into pinned host memory: map = next slot's device pointer, unmap =
// asynchronous device->host copy behind the frame that was rendered into the slot (ptrt_render
// marks the slot on its own stream, see include/ptrt.h), blit = wait for the OLDEST frame in flight
// and hand its host pixels (RGB8 bottom-up, exactly what the PBO held) to glTexSubImage2D, draw =
// the textured quad with the reference's flip.  With the default two slots the copy of frame i
// overlaps the rendering of frame i+1; the picture shown lags the render by slots-1 frames
// (PTRT_VIEW_SLOTS=1 in the environment: no lag, no overlap).
//
// Two builds of the same calls:
//   * GL (when <GLFW/glfw3.h> and <glad/gl.h> are on the include path, unless PTRT_VIEW_HEADLESS):
//     window, texture, shader and quad as in glfw_view_interop.hpp:174-332;
//   * headless (this image has no window system): draw_interop counts frames and can dump every
//     n-th one as a binary PPM (PTRT_VIEW_DUMP=<prefix>, PTRT_VIEW_DUMP_EVERY=<n>).
#pragma once
#include "scene.hpp"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>

#if !defined(PTRT_VIEW_HEADLESS) && defined(__has_include)
#if __has_include(<GLFW/glfw3.h>) && __has_include(<glad/gl.h>)
#define PTRT_VIEW_GL 1
#endif
#endif
#ifdef PTRT_VIEW_GL
#ifndef GLFW_INCLUDE_NONE
#define GLFW_INCLUDE_NONE
#endif
#include <glad/gl.h>
#include <GLFW/glfw3.h>
#endif

namespace rtgl {

struct InteropViewer {
#ifdef PTRT_VIEW_GL
    GLFWwindow *window = nullptr;
    GLuint tex = 0, vao = 0, vbo = 0, program = 0;
    int winX = 100, winY = 100, winW = 1280, winH = 720; // windowed placement, restored after fullscreen
    bool isFullscreen = false;
#endif
    int viewW = 0, viewH = 0; // size of the frames in the ring (for a band scene: its rows)
    ptrt_ring *ring = nullptr;
    int slots = 2;
    int mapped = -1; // slot handed out by map_pbo_device_ptr
    unsigned long long submitted = 0;
    // blit_pbo_to_texture / draw_interop take the viewer by const reference in the reference
    mutable unsigned long long presented = 0;
    mutable const unsigned char *host_frame = nullptr; // the frame blit exposed (NULL while the ring fills)
    std::string dump_prefix; // "" = no files; else <prefix>NNNNNN.ppm
    int dump_every = 0;
};

inline void check(int rc, const char *what) {
    if (rc != PTRT_OK)
        throw std::runtime_error(std::string(what) + ": " + ptrt_last_error(nullptr));
}

#ifdef PTRT_VIEW_GL
namespace detail {
// the image is stored bottom-up (scene.cuh:2013-2015) and GL's t axis points up, so the upload shows it
// upside-up only if t is mirrored -- the reference's `vUV = vec2(aUV.x, 1.0 - aUV.y)` (glfw_view_interop.hpp:159)
// applied to a quad whose aUV.y runs 1 (bottom) .. 0 (top): the picture's first byte row ends at the BOTTOM.
static const char *const kVertex = "#version 330 core\n"
                                   "layout(location = 0) in vec4 aPosUV;\n"
                                   "out vec2 vUV;\n"
                                   "void main() {\n"
                                   "    vUV = vec2(aPosUV.z, 1.0 - aPosUV.w);\n"
                                   "    gl_Position = vec4(aPosUV.xy, 0.0, 1.0);\n"
                                   "}\n";
static const char *const kFragment = "#version 330 core\n"
                                     "in vec2 vUV;\n"
                                     "out vec4 FragColor;\n"
                                     "uniform sampler2D uTex;\n"
                                     "void main() { FragColor = texture(uTex, vUV); }\n";

inline GLuint build_stage(GLenum kind, const char *text) {
    const GLuint sh = glCreateShader(kind);
    glShaderSource(sh, 1, &text, nullptr);
    glCompileShader(sh);
    GLint good = 0;
    glGetShaderiv(sh, GL_COMPILE_STATUS, &good);
    if (good)
        return sh;
    std::string log(4096, '\0');
    GLsizei n = 0;
    glGetShaderInfoLog(sh, (GLsizei)log.size(), &n, &log[0]);
    glDeleteShader(sh);
    throw std::runtime_error("Shader compile failed: " + log.substr(0, (size_t)n));
}
inline GLuint build_program() {
    const GLuint vs = build_stage(GL_VERTEX_SHADER, kVertex), fs = build_stage(GL_FRAGMENT_SHADER, kFragment);
    const GLuint prog = glCreateProgram();
    glAttachShader(prog, vs);
    glAttachShader(prog, fs);
    glLinkProgram(prog);
    glDeleteShader(vs);
    glDeleteShader(fs);
    GLint good = 0;
    glGetProgramiv(prog, GL_LINK_STATUS, &good);
    if (good)
        return prog;
    std::string log(4096, '\0');
    GLsizei n = 0;
    glGetProgramInfoLog(prog, (GLsizei)log.size(), &n, &log[0]);
    glDeleteProgram(prog);
    throw std::runtime_error("Program link failed: " + log.substr(0, (size_t)n));
}
inline void toggle_fullscreen(InteropViewer &V) { // F11, glfw_view_interop.hpp:77-105
    if (!V.isFullscreen) {
        glfwGetWindowPos(V.window, &V.winX, &V.winY);
        glfwGetWindowSize(V.window, &V.winW, &V.winH);
        GLFWmonitor *mon = glfwGetPrimaryMonitor();
        const GLFWvidmode *mode = glfwGetVideoMode(mon);
        glfwSetWindowMonitor(V.window, mon, 0, 0, mode->width, mode->height, mode->refreshRate);
    } else {
        glfwSetWindowMonitor(V.window, nullptr, V.winX, V.winY, V.winW, V.winH, 0);
    }
    V.isFullscreen = !V.isFullscreen;
}
// window + GL objects.  Unlike the reference the texture keeps the FRAME's size when the window is resized (the
// quad is stretched): the reference re-creates its PBO at the window's size while the scene keeps rendering
// width x height pixels into it.
inline void open_window(InteropViewer &V, const char *title) {
    if (!glfwInit())
        throw std::runtime_error("glfwInit failed");
    glfwWindowHint(GLFW_CONTEXT_VERSION_MAJOR, 3);
    glfwWindowHint(GLFW_CONTEXT_VERSION_MINOR, 3);
    glfwWindowHint(GLFW_OPENGL_PROFILE, GLFW_OPENGL_CORE_PROFILE);
    V.window = glfwCreateWindow(V.viewW, V.viewH, title ? title : "", nullptr, nullptr);
    if (!V.window)
        throw std::runtime_error("glfwCreateWindow failed");
    glfwMakeContextCurrent(V.window);
    glfwSwapInterval(0);
    if (!gladLoadGL(glfwGetProcAddress))
        throw std::runtime_error("Failed to load GL with GLAD (gladLoadGL)");
    glfwSetWindowUserPointer(V.window, &V);
    glfwSetKeyCallback(V.window, [](GLFWwindow *w, int key, int, int action, int) {
        if (action == GLFW_PRESS && key == GLFW_KEY_F11)
            toggle_fullscreen(*static_cast<InteropViewer *>(glfwGetWindowUserPointer(w)));
    });
    glGenTextures(1, &V.tex);
    glBindTexture(GL_TEXTURE_2D, V.tex);
    for (GLenum filter : {GL_TEXTURE_MIN_FILTER, GL_TEXTURE_MAG_FILTER})
        glTexParameteri(GL_TEXTURE_2D, filter, GL_NEAREST);
    for (GLenum wrap : {GL_TEXTURE_WRAP_S, GL_TEXTURE_WRAP_T})
        glTexParameteri(GL_TEXTURE_2D, wrap, GL_CLAMP_TO_EDGE);
    glPixelStorei(GL_UNPACK_ALIGNMENT, 1); // rows of 3*W bytes are not 4-aligned in general
    glTexImage2D(GL_TEXTURE_2D, 0, GL_RGB8, V.viewW, V.viewH, 0, GL_RGB, GL_UNSIGNED_BYTE, nullptr);
    // two triangles, {x, y, u, v}: v = 1 at the bottom edge, as in the reference's vertex table
    static const float quad[6][4] = {{-1, -1, 0, 1}, {1, -1, 1, 1}, {1, 1, 1, 0}, {-1, -1, 0, 1}, {1, 1, 1, 0}, {-1, 1, 0, 0}};
    glGenVertexArrays(1, &V.vao);
    glGenBuffers(1, &V.vbo);
    glBindVertexArray(V.vao);
    glBindBuffer(GL_ARRAY_BUFFER, V.vbo);
    glBufferData(GL_ARRAY_BUFFER, sizeof quad, quad, GL_STATIC_DRAW);
    glEnableVertexAttribArray(0);
    glVertexAttribPointer(0, 4, GL_FLOAT, GL_FALSE, 4 * sizeof(float), nullptr);
    V.program = build_program();
    glUseProgram(V.program);
    glUniform1i(glGetUniformLocation(V.program, "uTex"), 0);
}
inline void close_window(InteropViewer &V) {
    if (V.program)
        glDeleteProgram(V.program);
    if (V.vbo)
        glDeleteBuffers(1, &V.vbo);
    if (V.vao)
        glDeleteVertexArrays(1, &V.vao);
    if (V.tex)
        glDeleteTextures(1, &V.tex);
    if (V.window)
        glfwDestroyWindow(V.window);
    glfwTerminate();
}
} // namespace detail
#endif // PTRT_VIEW_GL

// init_interop_viewer(V, width, height, title, cudaDevice) -- glfw_view_interop.hpp:174
inline void init_interop_viewer(InteropViewer &V, int width, int height, const char *title, int device = 0) {
    if (width < 1 || height < 1)
        throw std::runtime_error("init_interop_viewer: bad size");
    V.viewW = width;
    V.viewH = height;
    if (const char *e = std::getenv("PTRT_VIEW_SLOTS"))
        V.slots = std::atoi(e);
    if (const char *e = std::getenv("PTRT_VIEW_DUMP"))
        V.dump_prefix = e;
    if (const char *e = std::getenv("PTRT_VIEW_DUMP_EVERY"))
        V.dump_every = std::atoi(e);
    V.mapped = -1;
    V.submitted = V.presented = 0;
    V.host_frame = nullptr;
    check(ptrt_ring_create(device, (size_t)width * height * 3, V.slots, &V.ring), "init_interop_viewer");
#ifdef PTRT_VIEW_GL
    detail::open_window(V, title);
#else
    (void)title;
#endif
}
// the same for a Scene that may be a band (tile) of a frame: the ring holds ITS rows on ITS device
inline void init_interop_viewer(InteropViewer &V, Scene &scene, const char *title = "", int slots = 2) {
    if (!scene.backend())
        throw std::runtime_error("init_interop_viewer: host-only Scene has no GPU back end");
    V.slots = slots;
    init_interop_viewer(V, scene.getWidth(), scene.getTileRows(), title, scene.getDevice());
}

inline uint8_t *map_pbo_device_ptr(InteropViewer &V, size_t *nbytes = nullptr) {
    void *p = nullptr;
    V.mapped = (int)(V.submitted % (unsigned long long)V.slots);
    check(ptrt_ring_map(V.ring, V.mapped, &p), "map_pbo_device_ptr");
    if (nbytes)
        *nbytes = (size_t)V.viewW * V.viewH * 3;
    return static_cast<uint8_t *>(p);
}

inline void unmap_pbo(InteropViewer &V) {
    if (V.mapped < 0)
        throw std::runtime_error("unmap_pbo: nothing mapped");
    check(ptrt_ring_unmap(V.ring, V.mapped), "unmap_pbo");
    V.mapped = -1;
    V.submitted++;
}

// The oldest frame in flight becomes V.host_frame -- and, in the GL build, the texture -- once the ring is full
// (or `flush` is set: drain the ring at the end of a run).
inline void blit_pbo_to_texture(const InteropViewer &V, bool flush = false) {
    V.host_frame = nullptr;
    const unsigned long long lag = flush ? 1ull : (unsigned long long)V.slots;
    if (V.submitted - V.presented < lag)
        return;
    const int slot = (int)(V.presented % (unsigned long long)V.slots);
    check(ptrt_ring_acquire(V.ring, slot, &V.host_frame), "blit_pbo_to_texture");
    V.presented++;
#ifdef PTRT_VIEW_GL
    glBindTexture(GL_TEXTURE_2D, V.tex);
    glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, V.viewW, V.viewH, GL_RGB, GL_UNSIGNED_BYTE, V.host_frame);
#endif
}

inline void draw_interop(const InteropViewer &V) {
#ifdef PTRT_VIEW_GL
    int fbW = 0, fbH = 0;
    glfwGetFramebufferSize(V.window, &fbW, &fbH);
    glViewport(0, 0, fbW, fbH);
    glClear(GL_COLOR_BUFFER_BIT);
    glActiveTexture(GL_TEXTURE0);
    glBindTexture(GL_TEXTURE_2D, V.tex);
    glUseProgram(V.program);
    glBindVertexArray(V.vao);
    glDrawArrays(GL_TRIANGLES, 0, 6);
    glfwSwapBuffers(V.window);
    glfwPollEvents();
#endif
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
    ptrt_ring_destroy(V.ring);
#ifdef PTRT_VIEW_GL
    detail::close_window(V);
#endif
    V = InteropViewer{};
}

} // namespace rtgl

// ptrt/farm.hpp -- a frame farmed over the GPUs of one node from ONE C++ process.
//
// The reference renders on one GPU.  An application built on the Scene mirror scales a frame by holding one Scene
// per GPU, each rendering a part of the rows -- contiguous bands, or interleaved 8-row strips when the frame's cost is
// uneven (sky above, geometry below) -- and a ptrt_farm (include/ptrt.h) that gathers the parts onto the first GPU:
// device-to-device there, RCCL send / receive over xGMI from the others.  The scene is replicated: `build` runs on
// every Scene, `forEach` applies a change (camera move, light, material) to all of them.
//
//     TileFarm farm(1920, 1080, {0, 1, 2, 3, 4, 5, 6, 7}, TileFarm::Strips, buildMyScene);
//     farm.uploadToGPU();
//     rtgl::init_interop_viewer(V, 1920, 1080, "farm", /*device*/ 0);
//     for (;;) { uint8_t *d = rtgl::map_pbo_device_ptr(V); farm.render_to_device(d); rtgl::unmap_pbo(V); ... }
#pragma once
#include "scene.hpp"

#include <chrono>
#include <functional>
#include <memory>
#include <string>
#include <vector>

class TileFarm {
  public:
    enum Layout { Bands, Strips };

    // one Scene per entry of `devices` (a device may appear more than once: several parts on one GPU)
    TileFarm(int w, int h, const std::vector<int> &devices, Layout layout, const std::function<void(Scene &)> &build)
        : width(w), height(h) {
        const int n = (int)devices.size();
        if (n < 1)
            throw std::runtime_error("TileFarm: no devices");
        for (int r = 0; r < n; ++r) {
            if (layout == Strips && n > 1) {
                parts.push_back(std::make_unique<Scene>(w, h, Scene::Interleave{r, n}, devices[(size_t)r]));
            } else { // bands of h / n rows, the last one takes the remainder
                const int base = h / n, y0 = r * base, rows = (r < n - 1) ? base : h - y0;
                parts.push_back(std::make_unique<Scene>(w, h, y0, rows, devices[(size_t)r]));
            }
            build(*parts.back());
            parts.back()->setDenoiserEnabled(false); // parts have no post chain (Scene::postFrameFromDevice does it on a gathered frame)
            parts.back()->setBloomEnabled(false);
            // (no start / stop events around a part's kernel: two driver calls per part and frame that nobody reads)
            ptrt_set_option(parts.back()->backend(), "time_kernels", 0);
        }
        std::vector<ptrt_ctx *> ctxs;
        for (auto &p : parts)
            ctxs.push_back(p->backend());
        if (ptrt_farm_create(ctxs.data(), n, &farm) != PTRT_OK)
            throw std::runtime_error(std::string("TileFarm: ") + ptrt_last_error(nullptr));
    }
    ~TileFarm() {
        ptrt_farm_destroy(farm); // before the contexts it refers to
        parts.clear();
    }
    TileFarm(const TileFarm &) = delete;
    TileFarm &operator=(const TileFarm &) = delete;

    size_t size() const { return parts.size(); }
    int getWidth() const { return width; }
    int getHeight() const { return height; }
    Scene &scene(size_t i) { return *parts[i]; }
    const char *transport() const { return ptrt_farm_transport(farm); }
    template <class F> void forEach(F f) {
        for (auto &p : parts)
            f(*p);
    }
    void initBlueNoise() { forEach([](Scene &s) { s.initBlueNoise(); }); }
    void uploadToGPU() { forEach([](Scene &s) { s.uploadToGPU(); }); }

    // One frame into `device_pixels` (W*H*3 bytes on the device of the first part, bottom-up like
    // Scene::render_to_device's); returns without synchronising.
    void render_to_device(unsigned char *device_pixels) { frame(device_pixels, 1); }
    void render_to_host(unsigned char *host_pixels) { frame(host_pixels, 0); }
    void sync() { check(ptrt_farm_sync(farm)); }

  private:
    int width, height;
    std::vector<std::unique_ptr<Scene>> parts;
    ptrt_farm *farm = nullptr;

    void check(int rc) {
        if (rc != PTRT_OK)
            throw std::runtime_error(std::string("TileFarm: ") + ptrt_last_error(nullptr));
    }
    // every part renders into its own image, asynchronously, on its device -- enqueued from one worker thread per part
    // (ptrt_farm_parallel), since a part's host work (dirty checks, ptrt_render: 20-50 us) in a row would be of the order of
    // an eighth of a frame on the GPU -- then the gather
    void frame(unsigned char *pixels, int is_device) {
        const auto t0 = std::chrono::steady_clock::now();
        errors.assign(parts.size(), std::string());
        // the parts on the presenting GPU write their rows straight into the device frame the gather assembles
        dev_frame = static_cast<unsigned char *>(ptrt_farm_device_frame(farm, pixels, is_device));
        check(ptrt_farm_parallel(farm, [](int i, void *u) {
            TileFarm *self = static_cast<TileFarm *>(u);
            try {
                Scene &p = *self->parts[(size_t)i];
                if (self->dev_frame && ptrt_farm_part_is_local(self->farm, i))
                    p.render_to_frame(self->dev_frame);
                else
                    p.render_to_device(nullptr);
            } catch (const std::exception &e) {
                self->errors[(size_t)i] = e.what();
            }
        }, this));
        for (const std::string &e : errors)
            if (!e.empty())
                throw std::runtime_error("TileFarm: " + e);
        check(ptrt_farm_gather(farm, pixels, is_device));
        host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    std::vector<std::string> errors;
    unsigned char *dev_frame = nullptr;
    double host_us = 0.0;

  public:
    // host time (us) of the calling thread inside the last render_to_device / render_to_host
    double hostMicroseconds() const { return host_us; }
    void setParallel(bool on) { check(ptrt_farm_set_option(farm, "parallel", on ? 1 : 0)); }
};

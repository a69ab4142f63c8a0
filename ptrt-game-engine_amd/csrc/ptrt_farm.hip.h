// ptrt_farm.hip.h -- the tile farm below the C ABI (ptrt_farm_*, include/ptrt.h): band / strip contexts of ONE process
// on the GPUs of one node, their RGB8 images gathered onto the presenting device.  Included by ptrt_capi.hip (it
// reads the contexts' tile geometry, device and stream).
//
// Frames are cut either into contiguous bands (ptrt_create) or into interleaved 8-row strips
// (ptrt_create_interleaved); the farm takes any set of contexts that tiles the frame exactly once.  A frame:
//   1. every context renders its rows into its own RGB8 buffer (ptrt_render with a NULL target: asynchronous, on
//      the context's stream, on its device);
//   2. contexts on the PRESENTING device (that of the first context) are copied device-to-device behind an event;
//      contexts on other devices send their image with ncclSend on their own stream, the presenting device posts the
//      matching ncclRecv on the farm's stream -- one grouped call per frame (RCCL over xGMI: point-to-point links into
//      the presenter, 0.78 MB per band at 1080p / 8 GPUs, 3.1 MB at 4K);
//   3. bands land in the frame where they belong; strips are scattered with one strided copy per context.
// No host synchronisation inside a frame; the next render of a context waits (on its stream) until its image has
// been taken.  RCCL is loaded on first use (dlopen), so the library has no link-time dependency on it and a
// single-GPU box never touches it.
// A second transport needs no RCCL at all (SURVEY 8(e) names it): "peer-copy" -- hipMemcpyPeerAsync of a remote context's
// image into the presenting device's staging buffer on the farm's stream, behind the context's `rendered` event, and the
// context's next render behind a `taken` event of the presenting device.  Chosen when librccl cannot be loaded or
// ncclCommInitAll fails, by PTRT_FARM_TRANSPORT=peer in the environment, or by ptrt_farm_set_option("transport", 1);
// ptrt_farm_transport() says which one a farm uses.  PEER PATH UNVERIFIED ON HARDWARE as well (one-GPU boxes only).
// RCCL PATH UNVERIFIED ON HARDWARE: every test so far ran on a one-GPU box, where the transport is "device-copy"; dlopen,
// ncclCommInitAll, the grouped send/receive on mixed streams (also two contexts on one remote device) and CommDestroy have
// never executed.  tests/test_farm_gpu.py takes the devices it finds, so its first run on a multi-GPU node is that test.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>

// One worker thread per context: a frame's parts are enqueued in parallel (a ptrt_render costs 20-50 us of host time; eight
// of them in a row would be the same order as an eighth of a frame on the GPU).  A worker spins for `spin_us` after its last
// job before it sleeps on its condition variable, so a render loop finds it awake (no futex round trip per frame) and an
// idle application costs nothing.
struct FarmWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<unsigned> posted{0}, done{0};
    std::atomic<bool> quit{false};
    void (*fn)(int, void *) = nullptr;
    void *user = nullptr;
    int index = 0;
    std::atomic<int> spin_us{2000};
    void run() {
        unsigned seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            while (posted.load(std::memory_order_acquire) == seen && !quit.load(std::memory_order_relaxed)) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us.load(std::memory_order_relaxed))) {
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return posted.load(std::memory_order_acquire) != seen || quit.load(); });
                    break;
                }
                __builtin_ia32_pause();
            }
            if (quit.load())
                return;
            seen = posted.load(std::memory_order_acquire);
            fn(index, user);
            done.store(seen, std::memory_order_release);
        }
    }
    void post(void (*f)(int, void *), void *u) {
        fn = f;
        user = u;
        {
            std::lock_guard<std::mutex> lk(m); // (pairs with the wait above: no lost wake-up)
            posted.fetch_add(1, std::memory_order_release);
        }
        cv.notify_one();
    }
    void wait() const {
        const unsigned want = posted.load(std::memory_order_relaxed);
        while (done.load(std::memory_order_acquire) != want)
            __builtin_ia32_pause();
    }
};

struct ptrt_farm {
    std::vector<std::unique_ptr<FarmWorker>> workers; // one per context, started on first use
    double host_us = 0.0;                           // host time inside the last ptrt_farm_render / ptrt_farm_parallel + gather
    std::vector<int> part_rc;
    std::vector<std::string> part_err;
    std::vector<ptrt_ctx *> band;
    std::vector<hipEvent_t> rendered, taken; // per context: image complete / image copied out
    std::vector<unsigned char *> staging;    // per context on another device: where its image is received
    std::vector<int> comm_rank;              // per context: rank of its device in `comms` (-1: presenting device)
    std::vector<ncclComm_t> comms;           // one per distinct device, [0] = presenting device
    std::vector<int> comm_dev;
    int W = 0, H = 0, device = 0;
    hipStream_t stream = nullptr;            // presenting device: receives, copies
    unsigned char *d_frame = nullptr;        // assembled frame when the caller's target is host memory
    bool primed = false;
    int parallel = 1, spin_us = 2000;
    std::string transport = "device-copy";
    bool remote = false, peer = false;       // contexts on other devices exist / they are fetched with hipMemcpyPeerAsync
    std::string rccl_error;                  // why RCCL is not available to this farm (empty: it is, or was never needed)
};

namespace {

std::mutex g_farm_mutex;
std::set<ptrt_farm *> g_farms;

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib)
                break;
        }
        if (!r.lib)
            return;
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
        r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");
        r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.GetErrorString;
    });
    return r;
}

bool farm_live(ptrt_farm *f) {
    std::lock_guard<std::mutex> lock(g_farm_mutex);
    return f && g_farms.count(f);
}

void farm_stop_workers(ptrt_farm *f) {
    for (auto &w : f->workers) {
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->quit.store(true);
        }
        w->cv.notify_one();
        if (w->th.joinable())
            w->th.join();
    }
    f->workers.clear();
}

void farm_free(ptrt_farm *f) {
    farm_stop_workers(f);
    // nothing of the last frame may still be in flight when the communicators go: the receives and copies on the farm's
    // stream, the sends on the contexts' streams
    if (f->stream) {
        (void)hipSetDevice(f->device);
        (void)hipStreamSynchronize(f->stream);
    }
    for (ptrt_ctx *c : f->band)
        if (ctx_live(c)) {
            (void)hipSetDevice(c->device);
            (void)hipStreamSynchronize(c->stream);
        }
    for (size_t i = 0; i < f->comms.size(); ++i)
        if (f->comms[i]) {
            (void)hipSetDevice(f->comm_dev[i]);
            (void)rccl().CommDestroy(f->comms[i]);
        }
    (void)hipSetDevice(f->device);
    if (f->stream) {
        (void)hipStreamSynchronize(f->stream);
        (void)hipStreamDestroy(f->stream);
    }
    for (unsigned char *p : f->staging)
        if (p)
            (void)hipFree(p);
    if (f->d_frame)
        (void)hipFree(f->d_frame);
    for (size_t i = 0; i < f->band.size(); ++i) {
        if (ctx_live(f->band[i]))
            (void)hipSetDevice(f->band[i]->device);
        if (i < f->rendered.size() && f->rendered[i])
            (void)hipEventDestroy(f->rendered[i]);
        if (i < f->taken.size() && f->taken[i])
            (void)hipEventDestroy(f->taken[i]);
    }
    delete f;
}

// Copies one context's image (device pointer on the presenting device) to its place in the frame, on `st`.
// A band is one block of rows; the strips of an interleaved context lie `period` strips apart.  Both images are
// bottom-up: frame byte row of view row y is H-1-y, a context's byte row of its local row yl is rows-1-yl.
hipError_t place_image(const ptrt_ctx *c, const unsigned char *src, unsigned char *frame, hipStream_t st) {
    const size_t row = (size_t)c->W * 3;
    if (c->il_period <= 1)
        return hipMemcpyAsync(frame + (size_t)(c->H - (c->y0 + c->rows)) * row, src, (size_t)c->rows * row, hipMemcpyDeviceToDevice, st);
    const int strips = (c->H + 7) / 8, last_rows = c->H - (strips - 1) * 8; // the frame's last strip may be short
    int n_local = 0;
    for (int t = c->il_phase; t < strips; t += c->il_period)
        ++n_local;
    const int t_last = c->il_phase + (n_local - 1) * c->il_period; // the context's last strip (top-down) = first in its image
    const bool owns_short = (t_last == strips - 1) && last_rows != 8;
    const unsigned char *s = src;
    int n_full = n_local;
    if (owns_short) { // frame rows [0, last_rows)
        const hipError_t e = hipMemcpyAsync(frame, s, (size_t)last_rows * row, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess)
            return e;
        s += (size_t)last_rows * row;
        --n_full;
    }
    if (n_full <= 0)
        return hipSuccess;
    // remaining strips, from the context's lowest (largest t) to its first: frame rows H-8(t+1) .. H-8t, ascending
    const int t_low = c->il_phase + (n_full - 1) * c->il_period;
    unsigned char *d = frame + (size_t)(c->H - 8 * (t_low + 1)) * row;
    return hipMemcpy2DAsync(d, (size_t)c->il_period * 8 * row, s, 8 * row, 8 * row, (size_t)n_full, hipMemcpyDeviceToDevice, st);
}

} // namespace

int ptrt_farm_create(ptrt_ctx *const *bands, int n_bands, ptrt_farm **out) {
    if (!out)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_create: out is NULL");
    *out = nullptr;
    if (!bands || n_bands < 1)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_create: no contexts");
    for (int i = 0; i < n_bands; ++i)
        if (!ctx_live(bands[i]))
            return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_create: context %d is not a live context", i);
    const int W = bands[0]->W, H = bands[0]->H;
    // the contexts must tile the frame exactly once
    std::vector<int> owner((size_t)H, -1);
    for (int i = 0; i < n_bands; ++i) {
        const ptrt_ctx *c = bands[i];
        if (c->W != W || c->H != H)
            return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_create: context %d renders a %dx%d frame, context 0 %dx%d", i, c->W, c->H, W, H);
        if (c->scaled())
            return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_create: context %d has a reduced render size", i);
        for (int yl = 0; yl < c->rows; ++yl) {
            const int y = c->il_period > 1 ? ((((yl >> 3) * c->il_period + c->il_phase) << 3) | (yl & 7)) : c->y0 + yl;
            if (y < 0 || y >= H || owner[(size_t)y] >= 0)
                return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_create: row %d is rendered by contexts %d and %d (or lies outside the frame)",
                            y, y >= 0 && y < H ? owner[(size_t)y] : -1, i);
            owner[(size_t)y] = i;
        }
    }
    for (int y = 0; y < H; ++y)
        if (owner[(size_t)y] < 0)
            return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_create: no context renders row %d", y);

    ptrt_farm *f = new ptrt_farm;
    f->band.assign(bands, bands + n_bands);
    f->W = W;
    f->H = H;
    f->device = bands[0]->device;
    f->rendered.assign((size_t)n_bands, nullptr);
    f->taken.assign((size_t)n_bands, nullptr);
    f->staging.assign((size_t)n_bands, nullptr);
    f->comm_rank.assign((size_t)n_bands, -1);
    auto bail = [&](int code, const std::string &msg) {
        farm_free(f);
        return fail(nullptr, code, "ptrt_farm_create: %s", msg.c_str());
    };
    hipError_t e = hipSetDevice(f->device);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking);
    if (e != hipSuccess)
        return bail(PTRT_E_HIP, hipGetErrorString(e));
    // devices other than the presenting one talk to it over RCCL: one communicator per distinct device
    // (PTRT_FARM_FORCE_REMOTE=1, a test hook: every context but the first is treated like one on another device -- an image of its
    // own, a staging buffer, the peer-copy transport -- even on the presenting device, so that a one-GPU box executes that path:
    // hipMemcpyPeerAsync with equal source and destination device, the rendered / taken events, the strided placement)
    const bool force_remote = getenv("PTRT_FARM_FORCE_REMOTE") != nullptr;
    f->comm_dev.push_back(f->device);
    for (int i = 0; i < n_bands; ++i) {
        const int d = bands[i]->device;
        if (d == f->device && !(force_remote && i > 0))
            continue;
        size_t k = 0;
        while (k < f->comm_dev.size() && f->comm_dev[k] != d)
            ++k;
        if (k == f->comm_dev.size())
            f->comm_dev.push_back(d);
        f->comm_rank[(size_t)i] = (int)k;
        f->remote = true;
    }
    if (f->remote) {
        const char *want = getenv("PTRT_FARM_TRANSPORT");
        const bool want_peer = force_remote || (want && std::string(want) == "peer");
        if (!want_peer) {
            if (!rccl().ok) {
                f->rccl_error = "librccl.so could not be loaded";
            } else {
                f->comms.assign(f->comm_dev.size(), nullptr);
                const ncclResult_t r = rccl().CommInitAll(f->comms.data(), (int)f->comm_dev.size(), f->comm_dev.data());
                if (r != ncclSuccess) {
                    f->rccl_error = std::string("ncclCommInitAll: ") + rccl().GetErrorString(r);
                    f->comms.clear();
                }
            }
        }
        // without RCCL (not wanted, not loadable, or its communicators did not come up) the images are fetched by peer copies
        f->peer = want_peer || f->comms.empty();
        f->transport = f->peer ? "peer-copy" : "rccl";
        if (f->peer) {
            (void)hipSetDevice(f->device);
            for (size_t k = 1; k < f->comm_dev.size(); ++k) { // direct xGMI reads where the platform allows (else the runtime stages)
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, f->device, f->comm_dev[k]) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(f->comm_dev[k], 0);
                (void)hipGetLastError();
            }
        }
    }
    for (int i = 0; i < n_bands; ++i) {
        e = hipSetDevice(bands[i]->device);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&f->rendered[(size_t)i], hipEventDisableTiming);
        // (`taken` is recorded on the farm's stream: an event of the PRESENTING device, also for a context elsewhere)
        if (e == hipSuccess)
            e = hipSetDevice(f->device);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&f->taken[(size_t)i], hipEventDisableTiming);
        if (e == hipSuccess)
            e = hipSetDevice(bands[i]->device);
        if (e == hipSuccess && f->comm_rank[(size_t)i] >= 0) {
            e = hipSetDevice(f->device);
            if (e == hipSuccess)
                e = hipMalloc((void **)&f->staging[(size_t)i], bands[i]->npix * 3);
        }
        if (e != hipSuccess)
            return bail(PTRT_E_HIP, hipGetErrorString(e));
    }
    {
        std::lock_guard<std::mutex> lock(g_farm_mutex);
        g_farms.insert(f);
    }
    *out = f;
    return PTRT_OK;
}

int ptrt_farm_bands(const ptrt_farm *f) { return farm_live(const_cast<ptrt_farm *>(f)) ? (int)f->band.size() : 0; }

// 1: part i sits on the presenting device and may render straight into the frame (PTRT_OUT_DEVICE_FRAME); 0: its image travels
int ptrt_farm_part_is_local(const ptrt_farm *f, int i) {
    return farm_live(const_cast<ptrt_farm *>(f)) && i >= 0 && i < (int)f->band.size() && f->comm_rank[(size_t)i] < 0 ? 1 : 0;
}

const char *ptrt_farm_transport(const ptrt_farm *f) { return farm_live(const_cast<ptrt_farm *>(f)) ? f->transport.c_str() : ""; }

// The device frame a gather into (out_rgb8, out_is_device) assembles: the caller's, or the farm's own for a host target.
void *ptrt_farm_device_frame(ptrt_farm *f, void *out_rgb8, int out_is_device) {
    if (!farm_live(f) || !out_rgb8)
        return nullptr;
    if (out_is_device)
        return out_rgb8;
    if (!f->d_frame) {
        if (hipSetDevice(f->device) != hipSuccess || hipMalloc((void **)&f->d_frame, (size_t)f->W * f->H * 3) != hipSuccess)
            return nullptr;
    }
    return f->d_frame;
}

// Gathers the contexts' current RGB8 images into the frame: a context on the presenting device that rendered straight
// into the frame (PTRT_OUT_DEVICE_FRAME) is only waited for; one that rendered into its own image (a NULL target) is copied.
int ptrt_farm_gather(ptrt_farm *f, void *out_rgb8, int out_is_device) {
    if (!farm_live(f) || !out_rgb8)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_gather: bad argument");
    for (ptrt_ctx *c : f->band)
        if (!ctx_live(c))
            return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_gather: a context of the farm has been destroyed");
    const size_t frame_bytes = (size_t)f->W * f->H * 3;
    unsigned char *frame = (unsigned char *)ptrt_farm_device_frame(f, out_rgb8, out_is_device);
    if (!frame)
        return fail(nullptr, PTRT_E_HIP, "ptrt_farm_gather: no device frame");
    HIP_TRY(nullptr, hipSetDevice(f->device));
    const size_t n = f->band.size();
    // remote images, peer-copy transport: fetched by the presenting device behind the context's render
    if (f->remote && f->peer) {
        for (size_t i = 0; i < n; ++i) {
            ptrt_ctx *c = f->band[i];
            if (f->comm_rank[i] < 0)
                continue;
            HIP_TRY(nullptr, hipSetDevice(c->device));
            HIP_TRY(nullptr, hipEventRecord(f->rendered[i], c->stream));
            HIP_TRY(nullptr, hipSetDevice(f->device));
            HIP_TRY(nullptr, hipStreamWaitEvent(f->stream, f->rendered[i], 0));
            HIP_TRY(nullptr, hipMemcpyPeerAsync(f->staging[i], f->device, c->d_rgb8, c->device, c->npix * 3, f->stream));
            HIP_TRY(nullptr, hipEventRecord(f->taken[i], f->stream)); // the context's next render may overwrite its image behind this
            HIP_TRY(nullptr, hipStreamWaitEvent(c->stream, f->taken[i], 0));
        }
    }
    // remote images, RCCL transport: sends on the contexts' streams (behind their render), receives on the farm's stream, one group
    if (f->remote && !f->peer) {
        ncclResult_t r = rccl().GroupStart();
        for (size_t i = 0; i < n && r == ncclSuccess; ++i) {
            ptrt_ctx *c = f->band[i];
            const int k = f->comm_rank[i];
            if (k < 0)
                continue;
            r = rccl().Send(c->d_rgb8, c->npix * 3, ncclUint8, 0, f->comms[(size_t)k], c->stream);
            if (r == ncclSuccess)
                r = rccl().Recv(f->staging[i], c->npix * 3, ncclUint8, k, f->comms[0], f->stream);
        }
        const ncclResult_t r2 = rccl().GroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess)
            return fail(nullptr, PTRT_E_HIP, "ptrt_farm_gather: RCCL: %s", rccl().GetErrorString(r != ncclSuccess ? r : r2));
    }
    for (size_t i = 0; i < n; ++i) {
        ptrt_ctx *c = f->band[i];
        const unsigned char *src = f->staging[i];
        if (f->comm_rank[i] < 0) { // on the presenting device (the current one): behind the context's render
            HIP_TRY(nullptr, hipEventRecord(f->rendered[i], c->stream));
            HIP_TRY(nullptr, hipStreamWaitEvent(f->stream, f->rendered[i], 0));
            src = c->d_rgb8;
        }
        if (f->comm_rank[i] < 0 && c->last_frame_target == (void *)frame)
            continue; // (its rows are already in the frame: the wait above is all the gather owes it)
        HIP_TRY(nullptr, place_image(c, src, frame, f->stream));
        if (f->comm_rank[i] < 0) { // its next render must not overwrite the image before it has been taken
            HIP_TRY(nullptr, hipEventRecord(f->taken[i], f->stream));
            HIP_TRY(nullptr, hipStreamWaitEvent(c->stream, f->taken[i], 0));
        }
    }
    if (!out_is_device) {
        HIP_TRY(nullptr, hipMemcpyAsync(out_rgb8, frame, frame_bytes, hipMemcpyDeviceToHost, f->stream));
        HIP_TRY(nullptr, hipStreamSynchronize(f->stream));
    } else {
        // a presentation-ring slot as the target (rtgl::map_pbo_device_ptr -> TileFarm::render_to_device -> unmap_pbo): the
        // slot's download must wait for the copies above, which run on the farm's NON-BLOCKING stream -- the ring's fallback
        // (an event on the NULL stream) does not order behind it
        ring_mark_rendered(out_rgb8, f->stream);
    }
    return PTRT_OK;
}

// fn(i, user) for every context i of the farm, each on its own worker thread, all at once; returns when all are back.
// What a frame's per-part host work goes through: ptrt_farm_render's ptrt_render calls, and TileFarm::frame's
// Scene::render_to_device calls (host/ptrt/farm.hpp).  fn must not throw.  parallel = 0 (ptrt_farm_set_option): in a row.
int ptrt_farm_parallel(ptrt_farm *f, void (*fn)(int, void *), void *user) {
    if (!farm_live(f) || !fn)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_parallel: bad argument");
    const int n = (int)f->band.size();
    if (!f->parallel || n == 1) {
        for (int i = 0; i < n; ++i)
            fn(i, user);
        return PTRT_OK;
    }
    if (f->workers.empty())
        for (int i = 0; i < n; ++i) {
            f->workers.emplace_back(new FarmWorker);
            FarmWorker *w = f->workers.back().get();
            w->index = i;
            w->spin_us.store(f->spin_us);
            w->th = std::thread([w] { w->run(); });
        }
    for (int i = 1; i < n; ++i)
        f->workers[(size_t)i]->post(fn, user);
    fn(0, user); // (the caller's thread takes the first part)
    for (int i = 1; i < n; ++i)
        f->workers[(size_t)i]->wait();
    return PTRT_OK;
}

// One frame: every context renders its rows (asynchronously, each on its device, enqueued in parallel), then the gather.
int ptrt_farm_render(ptrt_farm *f, int frame_index, int spp, int max_depth, void *out_rgb8, int out_is_device) {
    if (!farm_live(f))
        return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_render: bad farm");
    const auto t0 = std::chrono::steady_clock::now();
    // parts on the presenting device write their rows straight into the device frame (no image of their own to copy)
    void *dev_frame = out_rgb8 ? ptrt_farm_device_frame(f, out_rgb8, out_is_device) : nullptr;
    struct Job {
        ptrt_farm *f;
        int frame, spp, depth;
        void *dev_frame;
    } job{f, frame_index, spp, max_depth, dev_frame};
    f->part_rc.assign(f->band.size(), PTRT_OK);
    f->part_err.assign(f->band.size(), std::string());
    const int prc = ptrt_farm_parallel(f, [](int i, void *u) {
        Job *j = static_cast<Job *>(u);
        ptrt_ctx *c = j->f->band[(size_t)i];
        const bool local = j->f->comm_rank[(size_t)i] < 0 && j->dev_frame;
        const int rc = local ? ptrt_render(c, j->frame, j->spp, j->depth, j->dev_frame, PTRT_OUT_DEVICE_FRAME)
                             : ptrt_render(c, j->frame, j->spp, j->depth, nullptr, 0);
        j->f->part_rc[(size_t)i] = rc;
        if (rc != PTRT_OK)
            j->f->part_err[(size_t)i] = ptrt_last_error(c); // (the error text is per thread: carried back to the caller's)
    }, &job);
    if (prc != PTRT_OK)
        return prc;
    for (size_t i = 0; i < f->band.size(); ++i)
        if (f->part_rc[i] != PTRT_OK)
            return fail(nullptr, f->part_rc[i], "ptrt_farm_render: part %zu: %s", i, f->part_err[i].c_str());
    const int rc = ptrt_farm_gather(f, out_rgb8, out_is_device);
    f->host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// host time (us) the caller's thread spent inside the last ptrt_farm_render (enqueue of every part + the gather's calls)
double ptrt_farm_host_us(const ptrt_farm *f) { return farm_live(const_cast<ptrt_farm *>(f)) ? f->host_us : -1.0; }

// parallel 0|1 (default 1): enqueue the parts from one worker thread per context, or in a row from the caller's thread;
// spin_us: how long a worker polls for the next frame before it sleeps (default 2000)
int ptrt_farm_set_option(ptrt_farm *f, const char *name, long long value) {
    if (!farm_live(f) || !name)
        return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_set_option: bad argument");
    const std::string n(name);
    if (n == "parallel")
        f->parallel = value ? 1 : 0;
    else if (n == "spin_us") {
        if (value < 0 || value > 1000000)
            return fail(nullptr, PTRT_E_INVALID, "spin_us must be 0..1000000");
        f->spin_us = (int)value;
        for (auto &w : f->workers)
            w->spin_us.store((int)value);
    } else if (n == "transport") { // 0: RCCL send / receive for contexts on other devices (if its communicators came up); 1: peer copies
        if (value != 0 && value != 1)
            return fail(nullptr, PTRT_E_INVALID, "transport must be 0 (rccl) or 1 (peer-copy)");
        if (value == 0 && f->remote && f->comms.empty())
            return fail(nullptr, PTRT_E_HIP, "ptrt_farm_set_option: RCCL is not available to this farm (%s)",
                        f->rccl_error.empty() ? "peer-copy was chosen at creation" : f->rccl_error.c_str());
        f->peer = value == 1;
        if (f->remote)
            f->transport = f->peer ? "peer-copy" : "rccl";
    } else
        return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_set_option: unknown option '%s'", name);
    return PTRT_OK;
}

// waits until the last gathered frame is complete on the presenting device
int ptrt_farm_sync(ptrt_farm *f) {
    if (!farm_live(f))
        return fail(nullptr, PTRT_E_INVALID, "ptrt_farm_sync: bad farm");
    HIP_TRY(nullptr, hipSetDevice(f->device));
    HIP_TRY(nullptr, hipStreamSynchronize(f->stream));
    return PTRT_OK;
}

void ptrt_farm_destroy(ptrt_farm *f) {
    {
        std::lock_guard<std::mutex> lock(g_farm_mutex);
        if (!f || !g_farms.count(f))
            return;
        g_farms.erase(f);
    }
    farm_free(f);
}

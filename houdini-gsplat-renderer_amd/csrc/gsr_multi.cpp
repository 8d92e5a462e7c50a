// gsr_multi.cpp -- tile rows sharded over several GPUs, with the frame's ONE collective inside the library.
//
// The reference draws everything from Houdini's single draw thread in one process
// (/root/reference/gsplat_plugin/src/DM_GSplatHook.C:30-39), so the primary form is SINGLE-PROCESS:
//   gsr_multi_*   one caller thread drives G contexts (one per GPU).  Every rank has a worker thread that queues its frame
//                 (kernel launches are host work: eight ranks issued from one thread would take longer than the frames run),
//                 the caller's thread then issues the frame's one collective: the band images go to rank 0 over xGMI --
//                 ncclRecv x (G-1) on the root and one ncclSend per peer in a single ncclGroup on a communicator made by
//                 ncclCommInitAll.
//   gsr_comm_*    the same gather for ONE PROCESS PER GPU (torchrun-style launches): ncclCommInitRank from a unique id the
//                 launcher distributes; per frame a rank calls gsr_comm_render and nothing else.
// Streams.  A rank's kernels run on its EXEC stream, the collective on its TRANSFER stream, and bands are double-buffered,
// so frame f's gather overlaps frame f+1's kernels on every GPU (at eight ranks a band's kernels take about as long as the
// gather: serialised, the gather would halve the frame rate).  The caller's stream is touched twice per frame: an event at
// entry (the target buffer may still be read by earlier work) and a wait for the frame's "gathered" event.
// Band layout (GSR_OPT_SHARD_LAYOUT = 1, the default of gsr_multi): rank g owns a contiguous band of tile rows, which IS a
// contiguous block of rows of the final image -- the peers' bands are received straight into the caller's framebuffer
// (zero copy, no stitch kernel); only the root's own band is copied there.  Interleaved layout: bands are received back to
// back and de-interleaved by k_stitch_bands on the transfer stream.
// RCCL is loaded at run time (dlopen) the first time a communicator is needed, so single-GPU users never map it.
// Transport COPY (hipMemcpyPeerAsync / device-to-device copies ordered by events) exists so that the whole path -- shard,
// render, gather, stitch -- also runs with several contexts on ONE GPU (the 1-GPU test box), where RCCL refuses duplicate
// devices; it doubles as a fallback when librccl cannot be loaded.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <thread>
#include <vector>

#include "../../include/gsplat_hip.h"

// hooks into gsr_api.hip (hidden symbols of the same library)
int gsr_internal_frame_begin(gsr_context* c, const gsr_camera* cam, const float* depth, int depth_is_device, float* out_dev);
int gsr_internal_frame_finish(gsr_context* c);
int gsr_internal_frame_check(gsr_context* c, const gsr_camera* cam, const float* depth, int depth_is_device, float* out_dev);
void* gsr_internal_stream(gsr_context* c);
int gsr_internal_device(gsr_context* c);
int gsr_internal_shard_layout(gsr_context* c);
int gsr_internal_set_error(int code, const char* text);
int gsr_internal_stitch(gsr_context* c, const float* gathered, int count, int width, int height, float* out, void* stream);

namespace {

int fail(int code, const char* fmt, ...)
{
    char buf[480];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return gsr_internal_set_error(code, buf);
}

#define HIP_OK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? GSR_E_OOM : GSR_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---- RCCL, resolved at run time ---------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy that is already mapped (e.g. the one a host framework ships) is reused; otherwise the system one
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names)
            if ((r.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
        if (!r.handle)
            for (const char* nm : names)
                if ((r.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        if (!r.handle) return;
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name))
        SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommInitAll, "ncclCommInitAll");
        SYM(CommDestroy, "ncclCommDestroy"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
        SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GetErrorString, "ncclGetErrorString");
        SYM(CommCount, "ncclCommCount"); SYM(CommUserRank, "ncclCommUserRank");
#undef SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send &&
               r.Recv && r.GetErrorString && r.CommCount && r.CommUserRank;
    });
    return r;
}

#define NCCL_OK(expr)                                                                                 \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != ncclSuccess) return fail(GSR_E_COMM, "%s: %s", #expr, rccl().GetErrorString(r_));   \
    } while (0)

static_assert(sizeof(ncclUniqueId) == GSR_COMM_ID_BYTES, "gsplat_hip.h GSR_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

struct DevBuf {
    float* p = nullptr;
    size_t cap = 0;   // floats
    int device = 0;
    int ensure(int dev, size_t floats)
    {
        if (floats <= cap && dev == device && p) return GSR_OK;
        release();
        HIP_OK(hipSetDevice(dev));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&p), (floats ? floats : 1) * sizeof(float)));
        cap = floats;
        device = dev;
        return GSR_OK;
    }
    void release()
    {
        if (p) { (void)hipSetDevice(device); (void)hipFree(p); }
        p = nullptr; cap = 0;
    }
};

size_t band_floats(int width, int height, int count) { return (size_t)gsr_band_rows(height, 0, count) * (size_t)width * 4; }

// band layout: the pixel rows of rank g's band that exist in the image, and the image row its band starts at
int band_first_row(int height, int g, int count) { return g * gsr_band_rows(height, 0, count); }
int band_live_rows(int height, int g, int count)
{
    const long lo = (long)band_first_row(height, g, count);
    long hi = lo + (long)gsr_band_rows(height, 0, count);
    if (hi > height) hi = height;
    return hi > lo ? (int)(hi - lo) : 0;
}

// one event, made on its device, without timing unless asked
int make_event(int dev, hipEvent_t* ev, bool timing = false)
{
    HIP_OK(hipSetDevice(dev));
    HIP_OK(hipEventCreateWithFlags(ev, timing ? hipEventDefault : hipEventDisableTiming));
    return GSR_OK;
}

// "is the buffer free again?" -- asked on the host first (it normally has been for a whole frame), a stream wait otherwise
int wait_unless_done(hipStream_t s, hipEvent_t ev, bool recorded)
{
    if (!recorded) return GSR_OK;
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return GSR_OK;
    if (q != hipErrorNotReady) { (void)hipGetLastError(); }
    HIP_OK(hipStreamWaitEvent(s, ev, 0));
    return GSR_OK;
}

// ---- one worker thread per rank (gsr_multi) ------------------------------------------------------
// The caller posts a job to every worker and waits for all of them.  A worker spins for a short while after its last job
// (frames of an interactive viewport follow each other within a few hundred microseconds), then sleeps on a condition variable.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<uint64_t> posted{0}, done{0};
    std::atomic<bool> quit{false};
    std::function<int()> job;
    int rc = GSR_OK;
    char err[480] = "";
};

void worker_main(Worker* w, int device)
{
    (void)hipSetDevice(device);
    uint64_t seen = 0;
    for (;;) {
        // spin, then sleep
        const auto t0 = std::chrono::steady_clock::now();
        unsigned long spins = 0;
        while (w->posted.load(std::memory_order_acquire) == seen && !w->quit.load(std::memory_order_acquire)) {
            if ((++spins & 0x3ffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) {
                std::unique_lock<std::mutex> lk(w->mu);
                w->cv.wait(lk, [&] { return w->posted.load(std::memory_order_acquire) != seen || w->quit.load(std::memory_order_acquire); });
                break;
            }
            __builtin_ia32_pause();
        }
        if (w->quit.load(std::memory_order_acquire)) return;
        seen = w->posted.load(std::memory_order_acquire);
        w->rc = w->job();
        if (w->rc) std::snprintf(w->err, sizeof w->err, "%s", gsr_last_error());
        w->done.store(seen, std::memory_order_release);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// single process, several GPUs
struct gsr_multi {
    std::vector<gsr_context*> ctx;
    std::vector<int> dev;
    int transport = GSR_TRANSPORT_COPY;
    std::vector<ncclComm_t> comm;        // RCCL transport: one per rank (ncclCommInitAll)
    // streams: the caller's (root device; ours unless gsr_multi_set_stream gave one), and per rank exec + transfer
    hipStream_t user = nullptr, own_user = nullptr;
    std::vector<hipStream_t> exec, xfer;
    // double-buffered bands: rank g > 0 renders frame f into band[g][f & 1]; the root into gathered[f & 1] (slot 0 of the
    // G bands the interleaved layout receives back to back)
    std::vector<DevBuf> band[2];
    DevBuf gathered[2];
    std::vector<DevBuf> depth;           // rank g > 0: copy of a root-resident depth image
    std::vector<hipEvent_t> ev_band[2];  // rank g: "band of frame f complete" on exec[g]
    std::vector<hipEvent_t> ev_sent[2];  // rank g > 0: "band of frame f has left" on xfer[g] (RCCL transport)
    std::vector<char> sent_rec[2];
    hipEvent_t ev_user = nullptr;        // the caller's stream position at entry
    hipEvent_t ev_frame[2] = {nullptr, nullptr};   // "frame f gathered" on xfer[0] (timing-enabled: the gather is measured between ...
    hipEvent_t ev_t0[2] = {nullptr, nullptr};      // ... this one, recorded in front of the first receive, and ev_frame)
    bool frame_rec[2] = {false, false};
    DevBuf final_fb;                     // staging of a host target
    uint64_t frame = 0;
    int shape_sig[3] = {0, 0, -1};       // width, height, layout of the frames the buffers were sized for
    bool uploading = false;
    // gather timing (gsr_multi_gather_stats)
    bool time_gather = false;
    bool t0_rec[2] = {false, false};
    double gather_ms = 0.0;
    int64_t gather_n = 0;
    // workers
    bool threaded = false;
    std::vector<std::unique_ptr<Worker>> worker;
    uint64_t seq = 0;
};

static void stop_workers(gsr_multi* m)
{
    for (auto& w : m->worker) {
        if (!w) continue;
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->quit.store(true, std::memory_order_release);
        }
        w->cv.notify_one();
        if (w->th.joinable()) w->th.join();
    }
    m->worker.clear();
}

// run fn(g) for every rank: on the ranks' worker threads when there are any, else one after the other here
template <typename F>
static int for_each_rank(gsr_multi* m, F fn)
{
    const int G = (int)m->ctx.size();
    if (!m->threaded) {
        int rc = GSR_OK;
        for (int g = 0; g < G; ++g) {
            const int r = fn(g);
            if (r && !rc) rc = r;
        }
        return rc;
    }
    const uint64_t seq = ++m->seq;
    for (int g = 0; g < G; ++g) {
        Worker& w = *m->worker[g];
        w.job = [fn, g]() { return fn(g); };
        {
            std::lock_guard<std::mutex> lk(w.mu);
            w.posted.store(seq, std::memory_order_release);
        }
        w.cv.notify_one();
    }
    int rc = GSR_OK;
    for (int g = 0; g < G; ++g) {
        Worker& w = *m->worker[g];
        while (w.done.load(std::memory_order_acquire) != seq) __builtin_ia32_pause();
        if (w.rc && !rc) rc = gsr_internal_set_error(w.rc, w.err);
    }
    return rc;
}

static void harvest_gather_time(gsr_multi* m, int b)
{
    if (!m->t0_rec[b] || !m->frame_rec[b]) return;
    m->t0_rec[b] = false;
    float ms = 0.0f;
    if (hipEventSynchronize(m->ev_frame[b]) == hipSuccess && hipEventElapsedTime(&ms, m->ev_t0[b], m->ev_frame[b]) == hipSuccess) {
        m->gather_ms += ms;
        m->gather_n += 1;
    }
}

extern "C" int gsr_multi_synchronize(gsr_multi* m)
{
    if (!m) return fail(GSR_E_INVALID, "gsr_multi_synchronize: NULL");
    for (gsr_context* c : m->ctx) {
        int rc = gsr_synchronize(c);
        if (rc) return rc;
    }
    for (size_t g = 0; g < m->xfer.size(); ++g)
        if (m->xfer[g]) { HIP_OK(hipSetDevice(m->dev[g])); HIP_OK(hipStreamSynchronize(m->xfer[g])); }
    if (m->user) { HIP_OK(hipSetDevice(m->dev[0])); HIP_OK(hipStreamSynchronize(m->user)); }
    return GSR_OK;
}

extern "C" void gsr_multi_destroy(gsr_multi* m)
{
    if (!m) return;
    stop_workers(m);
    (void)gsr_multi_synchronize(m);
    if (m->transport == GSR_TRANSPORT_RCCL && rccl().ok)
        for (ncclComm_t c : m->comm)
            if (c) (void)rccl().CommDestroy(c);
    for (int b = 0; b < 2; ++b) {
        for (auto& x : m->band[b]) x.release();
        m->gathered[b].release();
        for (size_t g = 0; g < m->ev_band[b].size(); ++g)
            if (m->ev_band[b][g]) { (void)hipSetDevice(m->dev[g]); (void)hipEventDestroy(m->ev_band[b][g]); }
        for (size_t g = 0; g < m->ev_sent[b].size(); ++g)
            if (m->ev_sent[b][g]) { (void)hipSetDevice(m->dev[g]); (void)hipEventDestroy(m->ev_sent[b][g]); }
        if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
        if (m->ev_frame[b]) (void)hipEventDestroy(m->ev_frame[b]);
        if (m->ev_t0[b]) (void)hipEventDestroy(m->ev_t0[b]);
    }
    for (auto& b : m->depth) b.release();
    m->final_fb.release();
    if (m->ev_user) (void)hipEventDestroy(m->ev_user);
    for (gsr_context* c : m->ctx)
        if (c) gsr_destroy(c);
    for (size_t g = 0; g < m->exec.size(); ++g)
        if (m->exec[g]) { (void)hipSetDevice(m->dev[g]); (void)hipStreamDestroy(m->exec[g]); }
    for (size_t g = 0; g < m->xfer.size(); ++g)
        if (m->xfer[g]) { (void)hipSetDevice(m->dev[g]); (void)hipStreamDestroy(m->xfer[g]); }
    if (m->own_user) { (void)hipSetDevice(m->dev[0]); (void)hipStreamDestroy(m->own_user); }
    delete m;
}

extern "C" int gsr_multi_create(const int* devices, int count, int transport, gsr_multi** out)
{
    if (!out) return fail(GSR_E_INVALID, "gsr_multi_create: out is NULL");
    *out = nullptr;
    if (!devices || count < 1 || count > 64) return fail(GSR_E_INVALID, "gsr_multi_create: need 1..64 devices");
    if (transport != GSR_TRANSPORT_AUTO && transport != GSR_TRANSPORT_RCCL && transport != GSR_TRANSPORT_COPY)
        return fail(GSR_E_INVALID, "gsr_multi_create: unknown transport %d", transport);
    const bool distinct = std::set<int>(devices, devices + count).size() == (size_t)count;
    if (transport == GSR_TRANSPORT_RCCL && !distinct)
        return fail(GSR_E_INVALID, "gsr_multi_create: RCCL needs distinct devices (use GSR_TRANSPORT_COPY for several contexts on one GPU)");
    if (transport == GSR_TRANSPORT_AUTO) transport = (distinct && count > 1 && rccl().ok) ? GSR_TRANSPORT_RCCL : GSR_TRANSPORT_COPY;
    if (transport == GSR_TRANSPORT_RCCL && !rccl().ok) return fail(GSR_E_COMM, "gsr_multi_create: librccl could not be loaded");

    gsr_multi* m = new (std::nothrow) gsr_multi();
    if (!m) return fail(GSR_E_OOM, "gsr_multi_create: host allocation failed");
    m->transport = transport;
    m->dev.assign(devices, devices + count);
    m->ctx.assign(count, nullptr);
    m->comm.assign(count, nullptr);
    m->exec.assign(count, nullptr);
    m->xfer.assign(count, nullptr);
    m->depth.resize(count);
    for (int b = 0; b < 2; ++b) {
        m->band[b].resize(count);
        m->ev_band[b].assign(count, nullptr);
        m->ev_sent[b].assign(count, nullptr);
        m->sent_rec[b].assign(count, 0);
    }
    int rc = GSR_OK;
    for (int g = 0; g < count && !rc; ++g) {
        rc = gsr_create(devices[g], &m->ctx[g]);
        if (!rc) rc = gsr_set_row_shard(m->ctx[g], g, count);
        // contiguous bands: a rank keeps ~1/count of the splats, and its band is a block of rows of the final image
        if (!rc) rc = gsr_set_option(m->ctx[g], GSR_OPT_SHARD_LAYOUT, 1);
        if (!rc && count > 1) {
            if (hipSetDevice(devices[g]) != hipSuccess || hipStreamCreateWithFlags(&m->exec[g], hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&m->xfer[g], hipStreamNonBlocking) != hipSuccess)
                rc = fail(GSR_E_HIP, "gsr_multi_create: stream creation failed on device %d", devices[g]);
            if (!rc) rc = gsr_set_stream(m->ctx[g], m->exec[g]);
            for (int b = 0; b < 2 && !rc; ++b) {
                rc = make_event(devices[g], &m->ev_band[b][g]);
                if (!rc) rc = make_event(devices[g], &m->ev_sent[b][g]);
            }
        }
    }
    if (!rc) {
        if (hipSetDevice(devices[0]) != hipSuccess || hipStreamCreateWithFlags(&m->own_user, hipStreamNonBlocking) != hipSuccess)
            rc = fail(GSR_E_HIP, "gsr_multi_create: stream creation failed on device %d", devices[0]);
        m->user = m->own_user;
        if (!rc) rc = make_event(devices[0], &m->ev_user);
        for (int b = 0; b < 2 && !rc; ++b) {
            rc = make_event(devices[0], &m->ev_frame[b], true);
            if (!rc) rc = make_event(devices[0], &m->ev_t0[b], true);
        }
        if (!rc && count == 1) rc = gsr_set_stream(m->ctx[0], m->user);   // one rank: the plain path on the caller's stream
    }
    if (!rc && transport == GSR_TRANSPORT_COPY && distinct)
        for (int g = 1; g < count; ++g) {   // direct GPU-to-GPU copies over xGMI instead of a bounce through the host
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[0], devices[g]) == hipSuccess && can) {
                (void)hipSetDevice(devices[0]);
                (void)hipDeviceEnablePeerAccess(devices[g], 0);
                (void)hipGetLastError();   // "already enabled" is fine
            }
        }
    if (!rc && transport == GSR_TRANSPORT_RCCL) {
        ncclResult_t r = rccl().CommInitAll(m->comm.data(), count, devices);
        if (r != ncclSuccess) rc = fail(GSR_E_COMM, "ncclCommInitAll over %d GPUs: %s", count, rccl().GetErrorString(r));
    }
    // one worker thread per rank (GSR_MULTI_THREADS=0 in the environment: every rank is queued from the caller's thread)
    if (!rc && count > 1) {
        const char* e = std::getenv("GSR_MULTI_THREADS");
        m->threaded = !(e && std::atoi(e) == 0);
        if (m->threaded) {
            for (int g = 0; g < count; ++g) {
                m->worker.emplace_back(new (std::nothrow) Worker());
                if (!m->worker.back()) { rc = fail(GSR_E_OOM, "gsr_multi_create: host allocation failed"); break; }
                try {
                    m->worker.back()->th = std::thread(worker_main, m->worker.back().get(), devices[g]);
                } catch (...) {
                    rc = fail(GSR_E_HIP, "gsr_multi_create: could not start the worker thread of rank %d", g);
                    break;
                }
            }
        }
    }
    if (rc) { gsr_multi_destroy(m); return rc; }
    *out = m;
    return GSR_OK;
}

extern "C" int gsr_multi_count(gsr_multi* m) { return m ? (int)m->ctx.size() : 0; }
extern "C" gsr_context* gsr_multi_context(gsr_multi* m, int rank) { return (m && rank >= 0 && rank < (int)m->ctx.size()) ? m->ctx[rank] : nullptr; }
extern "C" int gsr_multi_transport(gsr_multi* m) { return m ? m->transport : 0; }

extern "C" int gsr_multi_comm_info(gsr_multi* m, int* ranks_out, int* nranks_out)
{
    if (!m) return fail(GSR_E_INVALID, "gsr_multi_comm_info: NULL");
    const int G = (int)m->ctx.size();
    if (nranks_out) *nranks_out = 0;
    for (int g = 0; g < G; ++g) {
        int r = -1, n = 0;
        if (m->transport == GSR_TRANSPORT_RCCL && m->comm[g]) {
            NCCL_OK(rccl().CommUserRank(m->comm[g], &r));
            NCCL_OK(rccl().CommCount(m->comm[g], &n));
        }
        if (ranks_out) ranks_out[g] = r;
        if (nranks_out && g == 0) *nranks_out = n;
    }
    return GSR_OK;
}

extern "C" int gsr_multi_set_stream(gsr_multi* m, void* stream)
{
    if (!m) return fail(GSR_E_INVALID, "gsr_multi_set_stream: NULL");
    int rc = gsr_multi_synchronize(m);
    if (rc) return rc;
    m->user = stream ? reinterpret_cast<hipStream_t>(stream) : m->own_user;
    if (m->ctx.size() == 1) return gsr_set_stream(m->ctx[0], m->user);
    return GSR_OK;
}

extern "C" int gsr_multi_set_option(gsr_multi* m, int option, int value)
{
    if (!m) return fail(GSR_E_INVALID, "gsr_multi_set_option: NULL");
    if (option == GSR_OPT_DEFERRED_CHECK && value) return fail(GSR_E_INVALID, "gsr_multi_set_option: frames are gathered, so every rank's pair count is checked before the gather");
    if (option == GSR_OPT_SHARD_LAYOUT) {   // frames in flight were rendered in the old layout
        int rc = gsr_multi_synchronize(m);
        if (rc) return rc;
    }
    for (gsr_context* c : m->ctx) {
        int rc = gsr_set_option(c, option, value);
        if (rc) return rc;
    }
    return GSR_OK;
}

extern "C" int gsr_multi_gather_stats(gsr_multi* m, int enable, double* ms_total, int64_t* gathers)
{
    if (!m) return fail(GSR_E_INVALID, "gsr_multi_gather_stats: NULL");
    int rc = gsr_multi_synchronize(m);
    if (rc) return rc;
    harvest_gather_time(m, 0);
    harvest_gather_time(m, 1);
    if (ms_total) *ms_total = m->gather_ms;
    if (gathers) *gathers = m->gather_n;
    if (enable >= 0) {
        if ((enable != 0) != m->time_gather) { m->gather_ms = 0.0; m->gather_n = 0; }
        m->time_gather = enable != 0;
    }
    return GSR_OK;
}

// ---- staging: the splat cloud is REPLICATED (0.8 GB of 288 GB at 6 M splats): every rank culls the clusters outside its
// band and projects the rest -- cheaper than exchanging projected records every frame (multi-GPU)
extern "C" int gsr_multi_upload_begin(gsr_multi* m, int64_t total, int has_sh, const float origin[3])
{
    if (!m) return fail(GSR_E_INVALID, "gsr_multi_upload_begin: NULL");
    int rc = gsr_multi_synchronize(m);
    if (rc) return rc;
    rc = for_each_rank(m, [=](int g) { return gsr_upload_begin(m->ctx[g], total, has_sh, origin); });
    if (rc) { (void)gsr_multi_upload_abort(m); return rc; }
    m->uploading = true;
    return GSR_OK;
}

extern "C" int gsr_multi_upload_append(gsr_multi* m, int64_t n, const float* P, const uint16_t* Cd, const float* alpha,
                                       const uint16_t* scale, const uint16_t* orient, const uint16_t* shx, const uint16_t* shy,
                                       const uint16_t* shz)
{
    if (!m || !m->uploading) return fail(GSR_E_INVALID, "gsr_multi_upload_append: no upload in progress");
    return for_each_rank(m, [=](int g) { return gsr_upload_append(m->ctx[g], n, P, Cd, alpha, scale, orient, shx, shy, shz); });
}

extern "C" int gsr_multi_upload_end(gsr_multi* m)
{
    if (!m || !m->uploading) return fail(GSR_E_INVALID, "gsr_multi_upload_end: no upload in progress");
    m->uploading = false;
    int rc = for_each_rank(m, [=](int g) { return gsr_upload_end(m->ctx[g]); });
    if (rc) { (void)gsr_multi_upload_abort(m); return rc; }
    return GSR_OK;
}

extern "C" int gsr_multi_upload_abort(gsr_multi* m)
{
    if (!m) return fail(GSR_E_INVALID, "gsr_multi_upload_abort: NULL");
    m->uploading = false;
    for (gsr_context* c : m->ctx) (void)gsr_upload_abort(c);
    return GSR_OK;
}

extern "C" int gsr_multi_upload(gsr_multi* m, int64_t n, const float* P, const uint16_t* Cd, const float* alpha,
                                const uint16_t* scale, const uint16_t* orient, const uint16_t* shx, const uint16_t* shy,
                                const uint16_t* shz, const float origin[3])
{
    int rc = gsr_multi_upload_begin(m, n, (shx && shy && shz) ? 1 : 0, origin);
    if (!rc) rc = gsr_multi_upload_append(m, n, P, Cd, alpha, scale, orient, shx, shy, shz);
    if (!rc) rc = gsr_multi_upload_end(m);
    if (rc && m) (void)gsr_multi_upload_abort(m);
    return rc;
}

// ---- per frame ---------------------------------------------------------------------------------
extern "C" int gsr_multi_render(gsr_multi* m, const gsr_camera* cam, float* rgba_out, int out_is_device)
{
    return gsr_multi_render_depth(m, cam, nullptr, 0, rgba_out, out_is_device);
}

extern "C" int gsr_multi_render_depth(gsr_multi* m, const gsr_camera* cam, const float* depth, int depth_is_device,
                                      float* rgba_out, int out_is_device)
{
    if (!m || !cam || !rgba_out) return fail(GSR_E_INVALID, "gsr_multi_render: NULL argument");
    const int G = (int)m->ctx.size();
    if (G == 1) return gsr_render_depth(m->ctx[0], cam, depth, depth_is_device, rgba_out, out_is_device);
    if (cam->width <= 0 || cam->height <= 0 || cam->width > GSR_MAX_DIM || cam->height > GSR_MAX_DIM)
        return fail(GSR_E_INVALID, "gsr_multi_render: bad framebuffer size %dx%d", cam->width, cam->height);
    const int W = cam->width, H = cam->height;
    const size_t bf = band_floats(W, H, G);
    const size_t npx = (size_t)W * H;
    const int b = (int)(m->frame & 1u);
    const bool bands = gsr_internal_shard_layout(m->ctx[0]) == 1;
    int rc;
    harvest_gather_time(m, b);
    {   // a frame of another shape: buffers are about to be regrown, and frames in flight still use them
        const int sig[3] = {W, H, bands ? 1 : 0};
        if (std::memcmp(sig, m->shape_sig, sizeof sig) != 0) {
            if ((rc = gsr_multi_synchronize(m))) return rc;
            std::memcpy(m->shape_sig, sig, sizeof sig);
        }
    }
    if ((rc = m->gathered[b].ensure(m->dev[0], bands ? bf : bf * G))) return rc;
    for (int g = 1; g < G; ++g)
        if ((rc = m->band[b][g].ensure(m->dev[g], bf))) return rc;
    float* target = rgba_out;
    if (!out_is_device) {
        if ((rc = m->final_fb.ensure(m->dev[0], npx * 4))) return rc;
        target = m->final_fb.p;
    }
    // the caller's stream position now: the target may still be read by what it queued before this call, and a depth image
    // on the root GPU was produced there
    HIP_OK(hipSetDevice(m->dev[0]));
    HIP_OK(hipEventRecord(m->ev_user, m->user));
    const bool dev_depth = depth && depth_is_device;
    if (dev_depth)
        for (int g = 1; g < G; ++g)
            if ((rc = m->depth[g].ensure(m->dev[g], npx))) return rc;

    // 1. every rank queues its frame (worker threads), looks at its pair count, checks -- and if need be repairs -- its band
    rc = for_each_rank(m, [=](int g) -> int {
        hipStream_t s = m->exec[g];
        HIP_OK(hipSetDevice(m->dev[g]));
        float* band = g == 0 ? m->gathered[b].p : m->band[b][g].p;
        // the buffer of frame f - 2 must have been gathered (root: read by its transfer stream; peer: sent / copied)
        int r;
        if (g == 0 || m->transport == GSR_TRANSPORT_COPY) r = wait_unless_done(s, m->ev_frame[b], m->frame_rec[b]);
        else r = wait_unless_done(s, m->ev_sent[b][g], m->sent_rec[b][g] != 0);
        if (r) return r;
        const float* d = depth;
        if (dev_depth) {
            HIP_OK(hipStreamWaitEvent(s, m->ev_user, 0));
            if (g > 0) {   // a depth image that lives on the root GPU is copied to the peers (each rank tests against the FULL image)
                if (m->dev[g] == m->dev[0]) HIP_OK(hipMemcpyAsync(m->depth[g].p, depth, npx * 4, hipMemcpyDeviceToDevice, s));
                else HIP_OK(hipMemcpyPeerAsync(m->depth[g].p, m->dev[g], depth, m->dev[0], npx * 4, s));
                d = m->depth[g].p;
            }
        }
        if ((r = gsr_internal_frame_begin(m->ctx[g], cam, d, depth_is_device, band))) { (void)gsr_internal_frame_finish(m->ctx[g]); return r; }
        if ((r = gsr_internal_frame_finish(m->ctx[g]))) return r;
        if ((r = gsr_internal_frame_check(m->ctx[g], cam, d, depth_is_device, band))) return r;
        HIP_OK(hipEventRecord(m->ev_band[b][g], s));
        return GSR_OK;
    });
    if (rc) return rc;

    // 2. the frame's one collective, on the transfer streams: bands -> root
    hipStream_t x0 = m->xfer[0];
    HIP_OK(hipSetDevice(m->dev[0]));
    HIP_OK(hipStreamWaitEvent(x0, m->ev_user, 0));
    if (m->time_gather) { HIP_OK(hipEventRecord(m->ev_t0[b], x0)); m->t0_rec[b] = true; }
    auto dst_of = [&](int g) { return bands ? target + (size_t)band_first_row(H, g, G) * W * 4 : m->gathered[b].p + (size_t)g * bf; };
    auto cnt_of = [&](int g) { return bands ? (size_t)band_live_rows(H, g, G) * W * 4 : bf; };
    if (m->transport == GSR_TRANSPORT_RCCL) {
        for (int g = 1; g < G; ++g) {
            HIP_OK(hipSetDevice(m->dev[g]));
            HIP_OK(hipStreamWaitEvent(m->xfer[g], m->ev_band[b][g], 0));
        }
        NCCL_OK(rccl().GroupStart());
        ncclResult_t r = ncclSuccess;
        for (int g = 1; g < G && r == ncclSuccess; ++g) {
            if (cnt_of(g) == 0) continue;
            r = rccl().Recv(dst_of(g), cnt_of(g), ncclFloat, g, m->comm[0], x0);
            if (r == ncclSuccess) r = rccl().Send(m->band[b][g].p, cnt_of(g), ncclFloat, 0, m->comm[g], m->xfer[g]);
        }
        const ncclResult_t re = rccl().GroupEnd();
        if (r != ncclSuccess || re != ncclSuccess)
            return fail(GSR_E_COMM, "band gather: %s", rccl().GetErrorString(r != ncclSuccess ? r : re));
        for (int g = 1; g < G; ++g) {
            HIP_OK(hipSetDevice(m->dev[g]));
            HIP_OK(hipEventRecord(m->ev_sent[b][g], m->xfer[g]));
            m->sent_rec[b][g] = 1;
        }
        HIP_OK(hipSetDevice(m->dev[0]));
    } else {
        for (int g = 1; g < G; ++g) {
            if (cnt_of(g) == 0) continue;
            HIP_OK(hipStreamWaitEvent(x0, m->ev_band[b][g], 0));
            if (m->dev[g] == m->dev[0]) HIP_OK(hipMemcpyAsync(dst_of(g), m->band[b][g].p, cnt_of(g) * 4, hipMemcpyDeviceToDevice, x0));
            else HIP_OK(hipMemcpyPeerAsync(dst_of(g), m->dev[0], m->band[b][g].p, m->dev[g], cnt_of(g) * 4, x0));
        }
    }
    // 3. the root's own band: copied into place (bands), or everything de-interleaved (interleaved rows)
    HIP_OK(hipStreamWaitEvent(x0, m->ev_band[b][0], 0));
    if (bands) {
        if (cnt_of(0)) HIP_OK(hipMemcpyAsync(target, m->gathered[b].p, cnt_of(0) * 4, hipMemcpyDeviceToDevice, x0));
    } else if ((rc = gsr_internal_stitch(m->ctx[0], m->gathered[b].p, G, W, H, target, x0))) return rc;
    HIP_OK(hipEventRecord(m->ev_frame[b], x0));
    m->frame_rec[b] = true;
    m->frame += 1;
    if (!out_is_device) {
        HIP_OK(hipMemcpyAsync(rgba_out, target, npx * 16, hipMemcpyDeviceToHost, x0));
        HIP_OK(hipStreamSynchronize(x0));
    }
    // the result is ordered on the caller's stream
    HIP_OK(hipStreamWaitEvent(m->user, m->ev_frame[b], 0));
    return GSR_OK;
}

extern "C" int gsr_multi_get_stats(gsr_multi* m, int rank, gsr_stats* out)
{
    if (!m || rank < 0 || rank >= (int)m->ctx.size()) return fail(GSR_E_INVALID, "gsr_multi_get_stats: bad rank");
    return gsr_get_stats(m->ctx[rank], out);
}

// ---------------------------------------------------------------------------------------------
// one process per GPU: the same gather on a communicator built from a unique id
namespace {
struct CommState {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, dev = 0;
    hipStream_t user = nullptr;          // the context's public stream as the caller set it
    hipStream_t exec = nullptr, xfer = nullptr;
    DevBuf band[2];                      // rank > 0: its band; root: the gather buffer (its own band first)
    hipEvent_t ev_band[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_user = nullptr;
    bool done_rec[2] = {false, false};
    uint64_t frame = 0;
    int shape_sig[3] = {0, 0, -1};
};
std::map<gsr_context*, CommState>& comm_table()
{
    static std::map<gsr_context*, CommState> t;
    return t;
}
std::mutex& comm_mutex()
{
    static std::mutex mu;
    return mu;
}
CommState* comm_find(gsr_context* ctx)
{
    std::lock_guard<std::mutex> lk(comm_mutex());
    auto it = comm_table().find(ctx);
    return it == comm_table().end() ? nullptr : &it->second;
}
void comm_free(CommState& st)
{
    (void)hipSetDevice(st.dev);
    if (st.xfer) (void)hipStreamSynchronize(st.xfer);
    if (st.comm) (void)rccl().CommDestroy(st.comm);
    for (int b = 0; b < 2; ++b) {
        st.band[b].release();
        if (st.ev_band[b]) (void)hipEventDestroy(st.ev_band[b]);
        if (st.ev_done[b]) (void)hipEventDestroy(st.ev_done[b]);
    }
    if (st.ev_user) (void)hipEventDestroy(st.ev_user);
    if (st.exec) (void)hipStreamDestroy(st.exec);
    if (st.xfer) (void)hipStreamDestroy(st.xfer);
}
}  // namespace

extern "C" int gsr_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int gsr_comm_get_unique_id(void* id)
{
    if (!id) return fail(GSR_E_INVALID, "gsr_comm_get_unique_id: NULL");
    if (!rccl().ok) return fail(GSR_E_COMM, "gsr_comm_get_unique_id: librccl could not be loaded");
    ncclUniqueId u;
    NCCL_OK(rccl().GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return GSR_OK;
}

// gsr_set_stream on a context with a communicator: the caller's stream is where results are ORDERED; the kernels keep
// their own stream (hook called by gsr_set_stream; true = handled)
__attribute__((visibility("hidden"))) bool gsr_internal_comm_set_user_stream(gsr_context* ctx, void* stream, void* own)
{
    CommState* st = comm_find(ctx);
    if (!st || st->world == 1) return false;
    (void)hipSetDevice(st->dev);
    (void)hipStreamSynchronize(st->xfer);
    st->user = stream ? reinterpret_cast<hipStream_t>(stream) : reinterpret_cast<hipStream_t>(own);
    return true;
}
// ... and gsr_synchronize: the transfer stream and the caller's stream too
__attribute__((visibility("hidden"))) int gsr_internal_comm_sync(gsr_context* ctx)
{
    CommState* st = comm_find(ctx);
    if (!st || st->world == 1) return GSR_OK;
    HIP_OK(hipSetDevice(st->dev));
    HIP_OK(hipStreamSynchronize(st->xfer));
    if (st->user) HIP_OK(hipStreamSynchronize(st->user));
    return GSR_OK;
}

extern "C" int gsr_comm_init(gsr_context* ctx, const void* id, int rank, int world)
{
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return fail(GSR_E_INVALID, "gsr_comm_init: bad argument");
    if (!rccl().ok) return fail(GSR_E_COMM, "gsr_comm_init: librccl could not be loaded");
    if (comm_find(ctx)) return fail(GSR_E_INVALID, "gsr_comm_init: context already has a communicator");
    CommState st;
    st.dev = gsr_internal_device(ctx);
    HIP_OK(hipSetDevice(st.dev));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    st.rank = rank;
    st.world = world;
    NCCL_OK(rccl().CommInitRank(&st.comm, world, u, rank));
    int rc = gsr_set_row_shard(ctx, rank, world);
    if (!rc && world > 1) {
        // the context's kernels move to a stream of their own: frame f's gather overlaps frame f + 1
        st.user = reinterpret_cast<hipStream_t>(gsr_internal_stream(ctx));
        if (hipStreamCreateWithFlags(&st.exec, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&st.xfer, hipStreamNonBlocking) != hipSuccess)
            rc = fail(GSR_E_HIP, "gsr_comm_init: stream creation failed");
        for (int b = 0; b < 2 && !rc; ++b) {
            rc = make_event(st.dev, &st.ev_band[b]);
            if (!rc) rc = make_event(st.dev, &st.ev_done[b]);
        }
        if (!rc) rc = make_event(st.dev, &st.ev_user);
        if (!rc) rc = gsr_set_stream(ctx, st.exec);
    }
    if (rc) { comm_free(st); (void)gsr_set_row_shard(ctx, 0, 1); return rc; }
    std::lock_guard<std::mutex> lk(comm_mutex());
    comm_table()[ctx] = st;
    return GSR_OK;
}

extern "C" int gsr_comm_info(gsr_context* ctx, int* rank, int* nranks)
{
    CommState* st = comm_find(ctx);
    if (!st) return fail(GSR_E_INVALID, "gsr_comm_info: gsr_comm_init first");
    int r = -1, n = 0;
    NCCL_OK(rccl().CommUserRank(st->comm, &r));
    NCCL_OK(rccl().CommCount(st->comm, &n));
    if (rank) *rank = r;
    if (nranks) *nranks = n;
    return GSR_OK;
}

static void comm_erase(gsr_context* ctx, bool restore_stream)
{
    CommState st;
    {
        std::lock_guard<std::mutex> lk(comm_mutex());
        auto it = comm_table().find(ctx);
        if (it == comm_table().end()) return;
        st = it->second;
        comm_table().erase(it);
    }
    if (restore_stream && st.world > 1) (void)gsr_set_stream(ctx, st.user);   // (the entry is gone: this reaches the context)
    comm_free(st);
}

extern "C" int gsr_comm_destroy(gsr_context* ctx)
{
    if (!comm_find(ctx)) return GSR_OK;
    (void)gsr_synchronize(ctx);
    comm_erase(ctx, true);
    (void)gsr_set_row_shard(ctx, 0, 1);
    return GSR_OK;
}

// called by gsr_destroy
__attribute__((visibility("hidden"))) void gsr_internal_comm_release(gsr_context* ctx) { comm_erase(ctx, false); }

extern "C" int gsr_comm_render(gsr_context* ctx, const gsr_camera* cam, const float* depth, int depth_is_device,
                               float* rgba_out_device)
{
    CommState* sp = comm_find(ctx);
    if (!sp) return fail(GSR_E_INVALID, "gsr_comm_render: gsr_comm_init first");
    CommState& st = *sp;
    if (!cam || (st.rank == 0 && !rgba_out_device)) return fail(GSR_E_INVALID, "gsr_comm_render: NULL argument");
    if (st.world == 1) return gsr_render_depth(ctx, cam, depth, depth_is_device, rgba_out_device, 1);
    if (cam->width <= 0 || cam->height <= 0 || cam->width > GSR_MAX_DIM || cam->height > GSR_MAX_DIM)
        return fail(GSR_E_INVALID, "gsr_comm_render: bad framebuffer size %dx%d", cam->width, cam->height);
    const int W = cam->width, H = cam->height, G = st.world;
    const size_t bf = band_floats(W, H, G);
    const bool bands = gsr_internal_shard_layout(ctx) == 1;
    const int b = (int)(st.frame & 1u);
    int rc;
    HIP_OK(hipSetDevice(st.dev));
    {   // a frame of another shape: the buffers are about to be regrown, and frames in flight still use them
        const int sig[3] = {W, H, bands ? 1 : 0};
        if (std::memcmp(sig, st.shape_sig, sizeof sig) != 0) {
            HIP_OK(hipStreamSynchronize(st.exec));
            HIP_OK(hipStreamSynchronize(st.xfer));
            std::memcpy(st.shape_sig, sig, sizeof sig);
        }
    }
    if ((rc = st.band[b].ensure(st.dev, (st.rank == 0 && !bands) ? bf * G : bf))) return rc;
    // the caller's stream position: what it queued before may still read the target, or have produced the depth image
    HIP_OK(hipEventRecord(st.ev_user, st.user));
    if ((rc = wait_unless_done(st.exec, st.ev_done[b], st.done_rec[b]))) return rc;    // frame f - 2 has left this buffer
    if (depth && depth_is_device) HIP_OK(hipStreamWaitEvent(st.exec, st.ev_user, 0));
    float* band = st.band[b].p;
    if ((rc = gsr_render_depth(ctx, cam, depth, depth_is_device, band, 1))) return rc;   // (on st.exec: the context's stream)
    HIP_OK(hipEventRecord(st.ev_band[b], st.exec));
    auto cnt_of = [&](int g) { return bands ? (size_t)band_live_rows(H, g, G) * W * 4 : bf; };
    if (st.rank == 0) {
        HIP_OK(hipStreamWaitEvent(st.xfer, st.ev_user, 0));
        NCCL_OK(rccl().GroupStart());
        ncclResult_t r = ncclSuccess;
        for (int g = 1; g < G && r == ncclSuccess; ++g) {
            if (cnt_of(g) == 0) continue;
            float* dst = bands ? rgba_out_device + (size_t)band_first_row(H, g, G) * W * 4 : st.band[b].p + (size_t)g * bf;
            r = rccl().Recv(dst, cnt_of(g), ncclFloat, g, st.comm, st.xfer);
        }
        const ncclResult_t re = rccl().GroupEnd();
        if (r != ncclSuccess || re != ncclSuccess) return fail(GSR_E_COMM, "band gather (root): %s", rccl().GetErrorString(r != ncclSuccess ? r : re));
        HIP_OK(hipStreamWaitEvent(st.xfer, st.ev_band[b], 0));
        if (bands) {
            if (cnt_of(0)) HIP_OK(hipMemcpyAsync(rgba_out_device, band, cnt_of(0) * 4, hipMemcpyDeviceToDevice, st.xfer));
        } else if ((rc = gsr_internal_stitch(ctx, st.band[b].p, G, W, H, rgba_out_device, st.xfer))) return rc;
    } else {
        HIP_OK(hipStreamWaitEvent(st.xfer, st.ev_band[b], 0));
        if (cnt_of(st.rank)) NCCL_OK(rccl().Send(band, cnt_of(st.rank), ncclFloat, 0, st.comm, st.xfer));
    }
    HIP_OK(hipEventRecord(st.ev_done[b], st.xfer));
    st.done_rec[b] = true;
    st.frame += 1;
    HIP_OK(hipStreamWaitEvent(st.user, st.ev_done[b], 0));
    return GSR_OK;
}

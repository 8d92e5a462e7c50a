// gsr_multi.cpp -- tile rows sharded over several GPUs, with the frame's ONE collective inside the library.
//
// The reference draws everything from Houdini's single draw thread in one process
// (/root/reference/gsplat_plugin/src/DM_GSplatHook.C:30-39), so the primary form is SINGLE-PROCESS:
//   gsr_multi_*   one caller thread drives G contexts (one per GPU, tile row r -> rank r % G).  A frame is QUEUED on
//                 every GPU before the host waits for any of them (gsr_internal_frame_begin / _finish), then the band
//                 images are gathered to rank 0 over xGMI -- ncclRecv x (G-1) on the root and one ncclSend per peer in
//                 a single ncclGroup on a communicator made by ncclCommInitAll -- and de-interleaved by k_stitch_bands.
//   gsr_comm_*    the same gather for ONE PROCESS PER GPU (torchrun-style launches): ncclCommInitRank from a unique id the
//                 launcher distributes; per frame a rank calls gsr_comm_render and nothing else.
// RCCL is loaded at run time (dlopen) the first time a communicator is needed, so single-GPU users never map it.
// Transport COPY (hipMemcpyPeerAsync / device-to-device copies ordered by events) exists so that the whole path -- shard,
// render, gather, stitch -- also runs with several contexts on ONE GPU (the 1-GPU test box), where RCCL refuses duplicate
// devices; it doubles as a fallback when librccl cannot be loaded.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <set>
#include <vector>

#include "../../include/gsplat_hip.h"

// hooks into gsr_api.hip (hidden symbols of the same library)
int gsr_internal_frame_begin(gsr_context* c, const gsr_camera* cam, const float* depth, int depth_is_device, float* out_dev);
int gsr_internal_frame_finish(gsr_context* c);
int gsr_internal_frame_check(gsr_context* c, const gsr_camera* cam, const float* depth, int depth_is_device, float* out_dev);
void* gsr_internal_stream(gsr_context* c);
int gsr_internal_device(gsr_context* c);
int gsr_internal_set_error(int code, const char* text);

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
#undef SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send &&
               r.Recv && r.GetErrorString;
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

}  // namespace

// ---------------------------------------------------------------------------------------------
// single process, several GPUs
struct gsr_multi {
    std::vector<gsr_context*> ctx;
    std::vector<int> dev;
    int transport = GSR_TRANSPORT_COPY;
    std::vector<ncclComm_t> comm;        // RCCL transport: one per rank (ncclCommInitAll)
    std::vector<DevBuf> band;            // rank g > 0: its band image on its own GPU
    std::vector<DevBuf> depth;           // rank g > 0: copy of a root-resident depth image
    std::vector<hipEvent_t> ev;          // rank g: "band complete" / root: "depth ready"
    DevBuf gathered, final_fb;           // root: G bands back to back (rank 0 renders straight into the first); full frame
    bool uploading = false;
};

extern "C" void gsr_multi_destroy(gsr_multi* m)
{
    if (!m) return;
    for (size_t g = 0; g < m->ctx.size(); ++g)
        if (m->ctx[g]) (void)gsr_synchronize(m->ctx[g]);
    if (m->transport == GSR_TRANSPORT_RCCL && rccl().ok)
        for (ncclComm_t c : m->comm)
            if (c) (void)rccl().CommDestroy(c);
    for (auto& b : m->band) b.release();
    for (auto& b : m->depth) b.release();
    m->gathered.release();
    m->final_fb.release();
    for (size_t g = 0; g < m->ev.size(); ++g)
        if (m->ev[g]) { (void)hipSetDevice(m->dev[g]); (void)hipEventDestroy(m->ev[g]); }
    for (gsr_context* c : m->ctx)
        if (c) gsr_destroy(c);
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
    m->band.resize(count);
    m->depth.resize(count);
    m->ev.assign(count, nullptr);
    int rc = GSR_OK;
    for (int g = 0; g < count && !rc; ++g) {
        rc = gsr_create(devices[g], &m->ctx[g]);
        if (!rc) rc = gsr_set_row_shard(m->ctx[g], g, count);
        if (!rc && (hipSetDevice(devices[g]) != hipSuccess || hipEventCreateWithFlags(&m->ev[g], hipEventDisableTiming) != hipSuccess))
            rc = fail(GSR_E_HIP, "gsr_multi_create: event creation failed on device %d", devices[g]);
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
    if (rc) { gsr_multi_destroy(m); return rc; }
    *out = m;
    return GSR_OK;
}

extern "C" int gsr_multi_count(gsr_multi* m) { return m ? (int)m->ctx.size() : 0; }
extern "C" gsr_context* gsr_multi_context(gsr_multi* m, int rank) { return (m && rank >= 0 && rank < (int)m->ctx.size()) ? m->ctx[rank] : nullptr; }
extern "C" int gsr_multi_transport(gsr_multi* m) { return m ? m->transport : 0; }

extern "C" int gsr_multi_set_stream(gsr_multi* m, void* stream)
{
    if (!m) return fail(GSR_E_INVALID, "gsr_multi_set_stream: NULL");
    return gsr_set_stream(m->ctx[0], stream);
}

extern "C" int gsr_multi_set_option(gsr_multi* m, int option, int value)
{
    if (!m) return fail(GSR_E_INVALID, "gsr_multi_set_option: NULL");
    if (option == GSR_OPT_DEFERRED_CHECK && value) return fail(GSR_E_INVALID, "gsr_multi_set_option: frames are gathered, so every rank's pair count is checked before the gather");
    for (gsr_context* c : m->ctx) {
        int rc = gsr_set_option(c, option, value);
        if (rc) return rc;
    }
    return GSR_OK;
}

// ---- staging: the splat cloud is REPLICATED (0.8 GB of 288 GB at 6 M splats): every rank projects all splats and keeps
// the ones that reach its tile rows -- cheaper than exchanging projected records every frame (multi-GPU)
extern "C" int gsr_multi_upload_begin(gsr_multi* m, int64_t total, int has_sh, const float origin[3])
{
    if (!m) return fail(GSR_E_INVALID, "gsr_multi_upload_begin: NULL");
    for (gsr_context* c : m->ctx) {
        int rc = gsr_upload_begin(c, total, has_sh, origin);
        if (rc) { (void)gsr_multi_upload_abort(m); return rc; }
    }
    m->uploading = true;
    return GSR_OK;
}

extern "C" int gsr_multi_upload_append(gsr_multi* m, int64_t n, const float* P, const uint16_t* Cd, const float* alpha,
                                       const uint16_t* scale, const uint16_t* orient, const uint16_t* shx, const uint16_t* shy,
                                       const uint16_t* shz)
{
    if (!m || !m->uploading) return fail(GSR_E_INVALID, "gsr_multi_upload_append: no upload in progress");
    for (gsr_context* c : m->ctx) {
        int rc = gsr_upload_append(c, n, P, Cd, alpha, scale, orient, shx, shy, shz);
        if (rc) return rc;
    }
    return GSR_OK;
}

extern "C" int gsr_multi_upload_end(gsr_multi* m)
{
    if (!m || !m->uploading) return fail(GSR_E_INVALID, "gsr_multi_upload_end: no upload in progress");
    m->uploading = false;
    for (gsr_context* c : m->ctx) {
        int rc = gsr_upload_end(c);
        if (rc) { (void)gsr_multi_upload_abort(m); return rc; }
    }
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
    const size_t bf = band_floats(cam->width, cam->height, G);
    const size_t npx = (size_t)cam->width * cam->height;
    int rc;
    if ((rc = m->gathered.ensure(m->dev[0], bf * G))) return rc;
    for (int g = 1; g < G; ++g)
        if ((rc = m->band[g].ensure(m->dev[g], bf))) return rc;
    float* target = rgba_out;
    if (!out_is_device) {
        if ((rc = m->final_fb.ensure(m->dev[0], npx * 4))) return rc;
        target = m->final_fb.p;
    }
    hipStream_t s0 = reinterpret_cast<hipStream_t>(gsr_internal_stream(m->ctx[0]));

    // a depth image that lives on the root GPU is copied to the peers (each rank tests against the FULL image)
    std::vector<const float*> dptr(G, depth);
    if (depth && depth_is_device) {
        HIP_OK(hipSetDevice(m->dev[0]));
        HIP_OK(hipEventRecord(m->ev[0], s0));
        for (int g = 1; g < G; ++g) {
            if ((rc = m->depth[g].ensure(m->dev[g], npx))) return rc;
            hipStream_t sg = reinterpret_cast<hipStream_t>(gsr_internal_stream(m->ctx[g]));
            HIP_OK(hipSetDevice(m->dev[g]));
            HIP_OK(hipStreamWaitEvent(sg, m->ev[0], 0));
            if (m->dev[g] == m->dev[0]) HIP_OK(hipMemcpyAsync(m->depth[g].p, depth, npx * 4, hipMemcpyDeviceToDevice, sg));
            else HIP_OK(hipMemcpyPeerAsync(m->depth[g].p, m->dev[g], depth, m->dev[0], npx * 4, sg));
            dptr[g] = m->depth[g].p;
        }
    }

    // 1. queue the frame on EVERY GPU (nothing below waits for a GPU) ...
    for (int g = 0; g < G; ++g) {
        float* band = g == 0 ? m->gathered.p : m->band[g].p;
        if ((rc = gsr_internal_frame_begin(m->ctx[g], cam, dptr[g], depth_is_device, band))) {
            for (int k = 0; k <= g; ++k) (void)gsr_internal_frame_finish(m->ctx[k]);
            return rc;
        }
    }
    // 2. ... then look at the pair counts (each GPU keeps working while the host reads the others')
    for (int g = 0; g < G; ++g) {
        const int r = gsr_internal_frame_finish(m->ctx[g]);
        if (r && !rc) rc = r;
    }
    if (rc) return rc;
    // (occlusion culling: a rank whose frame broke a depth horizon renders it again before its band travels)
    for (int g = 0; g < G; ++g) {
        float* band = g == 0 ? m->gathered.p : m->band[g].p;
        const int r = gsr_internal_frame_check(m->ctx[g], cam, dptr[g], depth_is_device, band);
        if (r && !rc) rc = r;
    }
    if (rc) return rc;

    // 3. the frame's one collective: bands -> root
    if (m->transport == GSR_TRANSPORT_RCCL) {
        NCCL_OK(rccl().GroupStart());
        ncclResult_t r = ncclSuccess;
        for (int g = 1; g < G && r == ncclSuccess; ++g) {
            r = rccl().Recv(m->gathered.p + (size_t)g * bf, bf, ncclFloat, g, m->comm[0], s0);
            if (r == ncclSuccess)
                r = rccl().Send(m->band[g].p, bf, ncclFloat, 0, m->comm[g], reinterpret_cast<hipStream_t>(gsr_internal_stream(m->ctx[g])));
        }
        const ncclResult_t re = rccl().GroupEnd();
        if (r != ncclSuccess || re != ncclSuccess)
            return fail(GSR_E_COMM, "band gather: %s", rccl().GetErrorString(r != ncclSuccess ? r : re));
    } else {
        for (int g = 1; g < G; ++g) {
            HIP_OK(hipSetDevice(m->dev[g]));
            HIP_OK(hipEventRecord(m->ev[g], reinterpret_cast<hipStream_t>(gsr_internal_stream(m->ctx[g]))));
        }
        HIP_OK(hipSetDevice(m->dev[0]));
        for (int g = 1; g < G; ++g) {
            HIP_OK(hipStreamWaitEvent(s0, m->ev[g], 0));
            float* dst = m->gathered.p + (size_t)g * bf;
            if (m->dev[g] == m->dev[0]) HIP_OK(hipMemcpyAsync(dst, m->band[g].p, bf * 4, hipMemcpyDeviceToDevice, s0));
            else HIP_OK(hipMemcpyPeerAsync(dst, m->dev[0], m->band[g].p, m->dev[g], bf * 4, s0));
        }
        // the peers' band buffers are read by the root's stream: a peer's NEXT frame must not overwrite its band before that
        // copy has run (the host returns after the pair counts, not after the copies, and a light rank finishes early)
        HIP_OK(hipEventRecord(m->ev[0], s0));
        for (int g = 1; g < G; ++g) {
            HIP_OK(hipSetDevice(m->dev[g]));
            HIP_OK(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(gsr_internal_stream(m->ctx[g])), m->ev[0], 0));
        }
        HIP_OK(hipSetDevice(m->dev[0]));
    }
    // 4. de-interleave on the root, on its public stream
    if ((rc = gsr_stitch_bands(m->ctx[0], m->gathered.p, G, cam->width, cam->height, target))) return rc;
    if (!out_is_device) {
        HIP_OK(hipSetDevice(m->dev[0]));
        HIP_OK(hipMemcpyAsync(rgba_out, target, npx * 16, hipMemcpyDeviceToHost, s0));
        HIP_OK(hipStreamSynchronize(s0));
    }
    return GSR_OK;
}

extern "C" int gsr_multi_synchronize(gsr_multi* m)
{
    if (!m) return fail(GSR_E_INVALID, "gsr_multi_synchronize: NULL");
    for (gsr_context* c : m->ctx) {
        int rc = gsr_synchronize(c);
        if (rc) return rc;
    }
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
    int rank = 0, world = 1;
    DevBuf band, gathered;
};
std::map<gsr_context*, CommState>& comm_table()
{
    static std::map<gsr_context*, CommState> t;
    return t;
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

extern "C" int gsr_comm_init(gsr_context* ctx, const void* id, int rank, int world)
{
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return fail(GSR_E_INVALID, "gsr_comm_init: bad argument");
    if (!rccl().ok) return fail(GSR_E_COMM, "gsr_comm_init: librccl could not be loaded");
    if (comm_table().count(ctx)) return fail(GSR_E_INVALID, "gsr_comm_init: context already has a communicator");
    HIP_OK(hipSetDevice(gsr_internal_device(ctx)));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    CommState st;
    st.rank = rank;
    st.world = world;
    NCCL_OK(rccl().CommInitRank(&st.comm, world, u, rank));
    int rc = gsr_set_row_shard(ctx, rank, world);
    if (rc) { (void)rccl().CommDestroy(st.comm); return rc; }
    comm_table()[ctx] = st;
    return GSR_OK;
}

extern "C" int gsr_comm_destroy(gsr_context* ctx)
{
    auto it = comm_table().find(ctx);
    if (it == comm_table().end()) return GSR_OK;
    (void)gsr_synchronize(ctx);
    if (it->second.comm) (void)rccl().CommDestroy(it->second.comm);
    it->second.band.release();
    it->second.gathered.release();
    comm_table().erase(it);
    (void)gsr_set_row_shard(ctx, 0, 1);
    return GSR_OK;
}

// called by gsr_destroy
__attribute__((visibility("hidden"))) void gsr_internal_comm_release(gsr_context* ctx)
{
    auto it = comm_table().find(ctx);
    if (it == comm_table().end()) return;
    if (it->second.comm) (void)rccl().CommDestroy(it->second.comm);
    it->second.band.release();
    it->second.gathered.release();
    comm_table().erase(it);
}

extern "C" int gsr_comm_render(gsr_context* ctx, const gsr_camera* cam, const float* depth, int depth_is_device,
                               float* rgba_out_device)
{
    auto it = comm_table().find(ctx);
    if (it == comm_table().end()) return fail(GSR_E_INVALID, "gsr_comm_render: gsr_comm_init first");
    CommState& st = it->second;
    if (!cam || (st.rank == 0 && !rgba_out_device)) return fail(GSR_E_INVALID, "gsr_comm_render: NULL argument");
    if (st.world == 1) return gsr_render_depth(ctx, cam, depth, depth_is_device, rgba_out_device, 1);
    const int dev = gsr_internal_device(ctx);
    const size_t bf = band_floats(cam->width, cam->height, st.world);
    int rc;
    float* band;
    if (st.rank == 0) {   // the root renders straight into slot 0 of the gather buffer
        if ((rc = st.gathered.ensure(dev, bf * st.world))) return rc;
        band = st.gathered.p;
    } else {
        if ((rc = st.band.ensure(dev, bf))) return rc;
        band = st.band.p;
    }
    if ((rc = gsr_render_depth(ctx, cam, depth, depth_is_device, band, 1))) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(gsr_internal_stream(ctx));   // the frame is ordered on it
    HIP_OK(hipSetDevice(dev));
    if (st.rank == 0) {
        NCCL_OK(rccl().GroupStart());
        ncclResult_t r = ncclSuccess;
        for (int g = 1; g < st.world && r == ncclSuccess; ++g) r = rccl().Recv(st.gathered.p + (size_t)g * bf, bf, ncclFloat, g, st.comm, s);
        const ncclResult_t re = rccl().GroupEnd();
        if (r != ncclSuccess || re != ncclSuccess) return fail(GSR_E_COMM, "band gather (root): %s", rccl().GetErrorString(r != ncclSuccess ? r : re));
        return gsr_stitch_bands(ctx, st.gathered.p, st.world, cam->width, cam->height, rgba_out_device);
    }
    NCCL_OK(rccl().Send(band, bf, ncclFloat, 0, st.comm, s));
    return GSR_OK;
}

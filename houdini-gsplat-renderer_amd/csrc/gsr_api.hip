// gsr_api.hip -- host side of libgsplat_hip.so: context, HBM buffers, the
// per-frame kernel pipeline and the C ABI declared in include/gsplat_hip.h.
//
// Pipeline per gsr_render (all on one HIP stream):
//   k_cluster_cull         clusters of 64 Morton-ordered splats vs clip planes / screen / band / depth horizons (k_cluster.h)
//   K1 k_preprocess        the splats of the surviving clusters -> record, key, (idx, rect)            (HBM)
//   depth sort             3 x {hist, row scan, scatter}; the FIRST pass drops culled splats,
//                          so everything downstream runs on the visible ones (count in HBM)
//   binning (k_binning.h)  counting sort of the (splat, super-tile) pairs into per-super-tile, depth-ordered lists:
//      k_bin_count         pairs per (block of depth-consecutive splats, super-tile)
//      k_scan_rows         per super-tile: exclusive scan over the blocks
//      k_bin_ranges        list ranges; pair count D -> mapped host memory
//      [stream sync: the host sizes the list buffer from D]
//      k_bin_place         every pair's (idx, rect) entry straight to its list position
//   K6 k_blend             per-tile list filtering + front-to-back compositing (VALU/LDS)
//   k_sum_work             per-tile bookkeeping -> counters -> mapped host memory
// Reference counterpart: GSplatRenderer::render + the GLSL program it drives
// (/root/reference/gsplat_plugin/src/GSplatRenderer.C:534-658).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <algorithm>
#include <new>
#include <vector>

#include "../../include/gsplat_hip.h"
#include "gsr_device.h"
#include "k_binning.h"
#include "k_blend.h"
#include "k_cluster.h"
#include "k_colour.h"
#include "k_preprocess.h"
#ifdef GSR_HOST_TIMING
static double g_t_verdict = 0, g_acc_py = 0, g_acc_pre = 0, g_acc_launch = 0; static long g_acc_n = 0;
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#endif
#include "k_sort.h"
#include "k_wire.h"
#include "gsr_policy.h"
static_assert(RS_SRC_BLOCK == GSR_K1_THREADS, "the gathering sort pass reads K1's per-workgroup compaction: 256 slots each");

#define GSR_VERSION_STR "gsplat_hip 0.1.0 (gfx950)"
#define GSR_MAX_SLOTS 2
#define GSR_STAGE_EVENTS 7
#define GSR_HOST_BANDS_MAX 8
#ifndef GSR_SPIN_US
#define GSR_SPIN_US 2000         // how long a host wait for a mailbox word spins before it blocks in the runtime
#endif

static thread_local char g_err[512] = "";

static int set_err(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return set_err(e_ == hipErrorOutOfMemory ? GSR_E_OOM : GSR_E_HIP, "%s failed: %s (%s:%d)", #expr, \
                           hipGetErrorString(e_), __FILE__, __LINE__);                             \
    } while (0)

// what identifies a depth order: the geometry generation, the shard (a rank sorts only the splats it owns) and the camera
struct SortKey {
    uint64_t gen = 0;
    int shard_index = 0, shard_count = 1, shard_layout = 0, flags = 0;
    gsr_camera cam{};
    bool same(const SortKey& o) const
    {
        return gen == o.gen && shard_index == o.shard_index && shard_count == o.shard_count && shard_layout == o.shard_layout && flags == o.flags &&
               std::memcmp(&cam, &o.cam, sizeof(gsr_camera)) == 0;
    }
};

// what a queued frame needs again when it is finished (or its back end re-queued)
struct FrameJob {
    bool open = false;
    GsrFrame f{};
    uint32_t n = 0;
    int local_tiles = 0, band_rows = 0, n_super = 0;
    size_t out_px = 0;
    float* target = nullptr;       // device buffer the blend kernel writes
    float* user_out = nullptr;     // the caller's pointer (host or device)
    const float* d_depth = nullptr;
    bool out_is_device = false, timing = false, timing_all = false, use_map = false;
    bool speculative = false;      // the back end was queued before the pair count was known
    int bn_items = 4;              // splats per thread of the binning kernels (k_binning.h): 4, or fewer for a small frame
    uint32_t bn_grid = 0;          // ... and their grid (workgroups that stride over the blocks)
    uint32_t k1_grid = 0;          // K1's grid: the estimate of its workgroup-iterations (sizes the bucket scatter's grid too)
    bool local_sort = false;       // the depth sort took the small-frame form (k_sort.h) ...
    bool sort_failed = false;      // ... and gave a bucket up: the frame is rendered again with the three global passes
    bool redo = false;             // phase 2 of a front-slab frame met a list buffer that was too short: the frame is rendered again (frame_check)
    bool ranges_folded = false;    // ... and k_bin_place forms the list ranges and posts the pair count itself (no k_bin_ranges launch)
    bool deferred = false;         // ... and the frame handed over without waiting for it (GSR_OPT_DEFERRED_CHECK)
    bool lazy = false;             // K1 left the SH colours pending (k_colour.h)
    bool cull = false;             // occlusion culling: k_cluster_cull and K1 drop what lies behind the slot's depth horizons
    bool direct = false;           // the frame runs on the public stream itself
    uint32_t ticket = 0;           // stamps the frame's pair count in the host mailbox
    // front-slab frames: 0 = an ordinary frame; 1 = the front slab (splats up to the slab key), followed at frame_finish by
    // 2 = the rest, culled against the tiles phase 1 left opaque.  The call's arguments are kept for queueing phase 2.
    int phase = 0;
    gsr_camera cam_arg{};
    const float* depth_arg = nullptr;
    int depth_is_device_arg = 0;
    // depth-tested frames: K1 (and k_cluster_cull) drop what lies behind everything the opaque pass left under the tiles it reaches
    bool dcull = false;            // ... against the slot's tile-max depth pyramid, parity dpar
    int dpar = 0;
    bool sort_fresh = false;       // this frame sorted (it did not reuse a cached order)
    bool dblind = false;           // ... and k_cluster_cull ran without the depth pyramids (they were built beside it, for K1)
    bool dstat = false;            // ... and the depth pyramid pass masked the covered depths with the tiles' status (a culled frame with a valid status)
    bool blend_guess_plain = false;   // the plain blend kernel was launched, guarded by "no pixel is covered" (queue_back_end)
    float* out_arg = nullptr;
    int out_is_device_arg = 0;
};

// Everything one frame in flight owns: its HIP stream, the per-frame HBM arrays, the small
// mailboxes and the stage events.  Two slots alternate, so that frame f+1's memory-bound front end
// (k_preprocess, depth sort, binning) overlaps frame f's VALU-bound k_blend on the GPU.
struct FrameSlot {
    hipStream_t own = nullptr;         // the slot's private stream
    hipStream_t stream = nullptr;      // the stream its frame runs on: `own` with two frames in flight; with strictly serial
                                       // frames the context's PUBLIC stream itself (no hand-over events at all)
    hipEvent_t ev_done = nullptr;      // end of the frame on `stream`
    hipEvent_t ev_user = nullptr;      // caller's stream position at gsr_render entry
    // host-target frames: the blend launch in bands of tile rows, each band's rows copied back while the next ones composite
    hipEvent_t ev_band[GSR_HOST_BANDS_MAX] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int bands = 0;                     // bands of the frame's last blend launch (0 / 1: one launch, one copy)
    int band_row[GSR_HOST_BANDS_MAX + 1] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipEvent_t ev_pairs = nullptr;     // the frame's pair count has reached host memory
    // per-splat frame buffers
    GsrRecord* rec = nullptr;
    uint32_t *keyA = nullptr, *keyB = nullptr;
    uint32_t* blk_cnt = nullptr;       // splats each K1 workgroup kept (their keys/payloads head the workgroup's 256 slots)
    uint2 *valA = nullptr, *valB = nullptr;      // depth-sort payload: (splat index, packed tile rect)
    uint32_t* d_n = nullptr;           // splats that survived culling = items after the first sort pass
    float* zwin = nullptr;             // per-splat window depth (depth-tested frames)
    float* depth_stage = nullptr;      // device copy of a host depth buffer
    size_t depth_cap = 0;
    float* dpyr = nullptr;             // depth-tested frames: the two tile-max pyramids of the opaque pass's depth (k_cluster.h), by
    size_t dpyr_cap = 0;               // frame parity: [par][which], dpyr_cap floats each
    uint32_t* dactive = nullptr;       // [2] by parity: "some tile's largest depth is below 1" (something can be culled)
    int dpar = 0;                      // parity of the slot's last depth-tested frame
    bool dpyr_built = false;           // the pyramids of parity `dpar` hold this frame's depth buffer (a frame that only LOOKED for covered pixels has none)
    bool sorted_dculled = false;       // the cached depth order holds a frame that was culled against its depth buffer
    // scan / sort scratch
    uint32_t* hist = nullptr;
    size_t hist_cap = 0;
    uint32_t* totals = nullptr;        // [512] per-digit totals of the current radix pass
    // pairs
    uint2* pvA = nullptr;                        // super-tile lists: (splat index, tile column/row mask) per entry
    size_t pair_cap = 0;
    int32_t *sstart = nullptr, *send = nullptr;  // super-tile ranges
    uint4* tile_work = nullptr;        // per tile: entries scanned, records gathered, wave-record evaluations
    size_t tile_cap = 0;
    // front-slab frames
    uint4* tile_work_a = nullptr;      // phase 1's per-tile bookkeeping (tile_cap entries)
    float* tbuf = nullptr;             // per pixel of the band: the transmittance phase 1 left
    size_t tbuf_cap = 0;
    float* hpyr2 = nullptr;            // the pyramid of "this tile is finished" phase 2 culls against (GSR_PYR_FLOATS)
    uint32_t* slab = nullptr;          // [GSR_SLAB_BINS + 8] histogram of the surviving clusters' nearest keys; then [0] the slab key, [1] clusters,
                                       // [2..5] the bucket ranges of the two phases' small-frame sorts (k_slab_pick)
    bool slab_dirty = false;           // a phase 1 filled the histogram and no phase 2 followed (its cull pass is what clears it): the
                                       // next front-slab frame clears it itself before it counts
    uint32_t slab_kept = 0;            // splats phase 1 sent to the depth sort
    uint32_t slab_kept1 = 0, slab_kept2 = 0;   // ... in this slot's LAST front-slab frame, per phase (0 = none yet): which sort a phase takes
    int32_t* redo = nullptr;           // lazy colour: tiles the plain blend kernel gave up (tile_cap entries)
    int32_t* order = nullptr;          // blockIdx -> tile, heaviest tiles first: written by k_tile_order at the end of a frame
    size_t order_cap = 0;              //   for this slot's next frame (valid while the tile geometry stays what it was)
    bool order_valid = false;
    int order_age = 0;                 // frames the table has stood (tiles alike: it is rebuilt every opt_order_keep frames only)
    int order_sig[6] = {0, 0, 0, 0, 0, 0}, order_per_xcd = 0;
    uint32_t* st_scan = nullptr;       // [512] per super-tile: deepest scan of its opaque tiles / "a tile stayed open" (k_tile_pass -> k_sum_work)
    GsrTilePartial* partial = nullptr; // [4096] per 8x8-tile block: partial sums of the frame's counters (k_tile_pass -> k_sum_work)
    uint32_t* sup_work = nullptr;      // [2][256] per-super-tile work sums of the blend kernel, by frame parity
    int sup_par = 0;
    // Depth horizons (occlusion culling): per TILE, the distance^2 beyond which this slot's NEXT frame needs nothing -- a max
    // pyramid written by k_sum_work at the end of every frame from how deep the frame's tiles had to look.  A culled frame is
    // checked there too (did a tile look past the horizon its splats were culled against?); the verdict goes to the mapped word
    // h_end, and the host renders a frame that failed again, without culling, before it hands it over.
    float* hpyr = nullptr;             // pyramid levels 0..3 (k_cluster.h), GSR_PYR_FLOATS floats: what the slot's next frame culls against
    float* hpyr_next = nullptr;        // ... double-buffered: a frame's last kernels read the one it was culled against while they fill the other
    float* hraw = nullptr;             // per-tile horizons before dilation (k_tile_pass -> k_horizon_dilate)
    float* hstat = nullptr;            // per tile: every tile of its neighbourhood was "classic" in the frame that left the horizons (k_blend.h: GsrHorizonArgs.stat)
    bool hstat_valid = false;          // ... written by a depth-tested frame's end (as old as the horizons)
    uint32_t* dbg_viol = nullptr;      // GSR_DEBUG_VIOL in the environment: which tiles broke their promise (k_tile_pass)
    int hpyr_re = 0;                   // the dilation radius built into hpyr
    // cluster culling (k_cluster.h): the ordered list of surviving clusters of the frame, as per-workgroup segments
    uint32_t* cseg = nullptr;          // [ngroups * per]
    uint32_t* ccnt = nullptr;          // [CC_MAX_GROUPS]
    uint32_t* d_counts = nullptr;      // [0] slots K1 filled, [1] surviving clusters, [2] the small-frame sort gave a bucket up
    uint32_t* bkt_key = nullptr;       // small-frame sort (k_sort.h): the bucket regions, BK_BUCKETS x BK_CAP keys ...
    uint2* bkt_val = nullptr;          // ... and payloads
    uint32_t* bkt_cnt = nullptr;       // ... and the bucket counters, BK_STRIDE apart
    uint32_t surv_hint = 0;            // surviving clusters of this slot's last frame (sizes K1's grid; 0 = unknown)
    uint32_t kept_hint = 0;            // splats that reached the depth sort in this slot's last frame (picks the sort; 0 = unknown)
    uint32_t kept_lo = 0, kept_hi = 0; // ... and the smallest / largest of their keys, as float bits of the distance^2 (0, 0 = unknown)
    bool kept_culled = false;          // ... in a frame that was occlusion-culled (an unculled one keeps ten times as much: no prediction across)
    GsrLocalSortPolicy local_pol;      // back-off of the small-frame sort (gsr_policy.h): a run of > 64 equal keys defeats it EVERY frame
    unsigned long long* h_end = nullptr;      // pinned + mapped: ticket << 32 | violation
    unsigned long long* h_end_dev = nullptr;
    bool horizon_valid = false;
    int horizon_sig[7] = {0, 0, 0, 0, 0, 0, 0};   // tile geometry + geometry generation the horizons belong to
    gsr_camera horizon_cam{};                     // ... and the camera of the frame that left them (camera_jumped)
    bool sorted_culled = false;        // the cached depth order holds a culled frame's splats only
    unsigned long long* lazy_ctr = nullptr;   // [0] low word: redo count of the frame, [1]: colours evaluated (running)
    uint32_t* colour_evals = nullptr;         // [256] colours evaluated per super-tile list (this frame; folded into lazy_ctr[1])
    bool last_lazy = false;            // the last frame of this slot left colours pending
    float* fb = nullptr;               // staging for host-pointer output
    size_t fb_cap = 0;
    int fb_sig[5] = {-1, -1, -1, -1, -1};   // the band shape the staging buffer was last cleared for
    // small device/host mailboxes
    unsigned long long* counters = nullptr;  // k_sum_work's layout: [1]/[2] records gathered (frame/running), [3]/[4] list entries
                                             // scanned, [5] wave-record evaluations (running)
    unsigned long long* h_total = nullptr;      // pinned + mapped + coherent [4]: k_bin_ranges writes (frame ticket << 32 | pair count), hints, survivors
    unsigned long long* h_total_dev = nullptr;  // its device-side address
    uint32_t ticket = 0;                        // ticket of the frame queued last in this slot
    unsigned long long* h_counters = nullptr;  // host copy of the frame's bookkeeping (fetched from d_frame when stats are asked for)
    unsigned long long* d_frame = nullptr;     // device: k_sum_work's per-frame summary [8]
    // depth-sort cache: the frame description whose order is in (keyA, valA)
    bool sort_valid = false;
    SortKey sort_key{};
    uint32_t key_min = 0;              // of the frame whose order is cached
    FrameJob job;                      // the frame queued in this slot (open until frame_finish)
    // last frame rendered in this slot
    int last_tiles_x = 0, last_local_ty = 0, last_supers = 0;
    uint32_t last_pairs = 0;
    int super_tile = 0, stiles_x = 0, stiles_y = 0;
    uint64_t frame_id = 0;             // 1-based id of that frame, 0 = never used
    hipEvent_t ev[GSR_STAGE_EVENTS];
    bool ev_pending = false;
    bool ev_all = false;             // all seven stage events were recorded (timing level 2), not just the blend kernel's
    bool ev_ok = false;
};

struct gsr_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;      // the PUBLIC stream: results are ordered on it

    // geometry (SoA of 16-byte vectors), shared read-only by the frame slots
    uint32_t n = 0, cap = 0;
    bool has_sh = false;
    float origin[3] = {0, 0, 0};
    uint64_t geo_gen = 0;              // generation of the resident geometry: only ever counts up (cached orders, horizons, the wire overlay's
                                       // inverse permutation are stamped with it), so a stamp of an older cloud can never match a newer one
    bool has_geometry = false;         // a complete upload is resident (gsr_render: GSR_E_NO_GEOMETRY otherwise)
    float4* geoA = nullptr;
    uint4* geoB = nullptr;
    uint4* col = nullptr;              // colour halves as SoA chunks (eager colour in K1)
    uint4* colrow = nullptr;           // ... and as one contiguous row per splat (lazy colour gathers by index)
    int col_chunks = 0;
    // spatially ordered storage (k_cluster.h): storage slot j holds the perm[j]-th splat of the upload; clusters of 64 slots
    uint32_t* perm = nullptr;          // NULL = upload order
    std::vector<int32_t> h_perm;       // host copy (debug read-backs un-permute through it), fetched on demand
    float4 *clusA = nullptr, *clusB = nullptr;
    uint32_t nclus = 0;
    uint32_t* prefix = nullptr;        // lazy colour: per super-tile, list entries to colour next frame (k_sum_work writes it)
    uint32_t* prefix_all = nullptr;    // 256 x 0xffffffff: colour every list completely (first frame, debug read-back)
    uint32_t* prefix_none = nullptr;   // 256 x 0: colour nothing ahead of time (GSR_FLAG_LAZY_NO_PREFIX: every tile takes the fallback)
    bool prefix_valid = false;
    bool uploading = false;
    uint32_t up_total = 0, up_filled = 0;
    // bounding box of the uploaded positions (bounds the sort keys of a frame)
    bool bbox_ok = false;
    double bb_lo[3] = {0, 0, 0}, bb_hi[3] = {0, 0, 0};

    FrameSlot slot[GSR_MAX_SLOTS];
    // position-keyed order (GSR_OPT_SORT_CACHE = 2): all splats in depth order for one camera position
    uint32_t* pos_order = nullptr;
    size_t pos_cap = 0;
    bool pos_valid = false;
    uint64_t pos_gen = 0;
    float pos_cam[3] = {0, 0, 0}, last_cam[3] = {0, 0, 0};
    uint32_t pos_kmin = 0, pos_kmax = 0;
    bool last_cam_set = false;
    uint32_t* blk_pre = nullptr;       // exclusive prefix of K1's per-iteration counts (k_scan_counts)
    size_t blk_pre_cap = 0;
    int opt_order_keep = 32;           // (A/B hook, GSR_ORDER_KEEP in the environment: 0 = no tile order where the tiles are alike)
    int opt_fuse_order = 1;            // (A/B hook, GSR_FUSE_ORDER) the tile order is built in the frame-end launch instead of behind it
    int opt_mid_sort = 1;              // (A/B hook, GSR_MID_SORT) RS_ITEMS_MID keys per thread in the global sort passes of mid-size frames
    int opt_host_bands = 4;            // (A/B hook, GSR_HOST_BANDS) host-target frames: bands of tile rows whose copy-back overlaps the compositing of the next (1: off)
    hipStream_t copy_stream = nullptr; // ... and the stream the copies run on
    int opt_k1_scatter = 1;            // (A/B hook, GSR_K1_SCATTER) the small-frame sort's bucket pass inside K1 (0: a kernel of its own behind it)
    int opt_scatter_direct = 1;        // (A/B hook, GSR_SCATTER_DIRECT in the environment) the small-frame sort's scatter: one workgroup per K1 block
    int opt_bn_items = 0;              // (A/B hook, GSR_BN_ITEMS in the environment: 1, 2 or 4 splats per binning thread; 0 = by frame size)
    int nslots = 1;                    // frames in flight (GSR_OPT_FRAMES_IN_FLIGHT): serial by default -- occlusion culling wants the
                                       // horizons of the frame just before, and two slots hand it those of the frame before that

    int32_t* tile_map = nullptr;       // blockIdx -> tile (XCD-aware order), -1 = idle block
    size_t map_cap = 0;
    int map_w = 0, map_h = 0, map_si = -1, map_sc = 0, map_rpb = -1, map_shift = -1, map_grid = 0;

    int shard_index = 0, shard_count = 1, shard_layout = 0;   // layout: 0 = interleaved rows, 1 = contiguous bands
    int opt_swizzle = 2, opt_timing = 1, opt_sort_cache = 1, opt_super = 0, opt_flags = 0, opt_deferred = 0, opt_lazy = 1, opt_cull = 1, opt_timing_every = 1;
    int opt_cluster = 1, opt_morton = 1, opt_local_sort = 1;
    int opt_slab = 1;                  // front-slab frames (GSR_OPT_FRONT_SLAB): 0 off, 1 where occlusion culling pays but has no horizons, 2 always
    int slab_frac = 40, slab_min = 4096, slab_max = 8192;   // the slab: this many 256ths of the surviving clusters, at least / at most so many
                                       // clusters (A/B hooks: GSR_SLAB_FRAC, GSR_SLAB_MIN, GSR_SLAB_MAX)
    bool classic_once = false;         // the next frame sorts with the three global passes whatever the prediction says

    gsr_stats st{};
    uint64_t frame_no = 0;
    size_t pair_want = 0;              // largest list buffer any frame slot needed so far
    int64_t lazy_base = 0;             // lazy_colours_total at the last gsr_stats_reset
    uint32_t* lazy_hint = nullptr;     // device: would lazy colour pay? (k_sum_work -> k_bin_ranges -> mailbox -> lazy_pays)
    bool lazy_pays = false;
    GsrCullPolicy cull_pol;            // occlusion culling: pays / weak / hold-off / back-off / dilation radius (gsr_policy.h; DESIGN.md section 4's state table)
    GsrSlabPolicy slab_pol;            // front-slab frames: held off for 256 frames after one that kept more than a third of an unculled frame (B1 at 0.46 loses 9 %)
    bool prefix_cheaper = false;       // the list-prefix colour pass would evaluate fewer colours than one per kept splat
    bool depth_active = false;         // the last depth-tested frame's depth buffer held opaque geometry in front of the far plane: the next one
                                       // builds its depth pyramid in a launch of its own, IN FRONT of k_cluster_cull (which then culls against it
                                       // too); otherwise the pyramid is built beside the cluster tests, for K1 alone (no extra launch)
    bool order_pays = false;           // k_sum_work's other verdict: the tiles differ enough in work for k_tile_order to pay
    unsigned long long* wire_zbuf = nullptr;   // wireframe overlay: (depth bits, splat index) per pixel ...
    float* wire_out = nullptr;                 // ... and the image staged for a host target
    uint32_t* wire_inv = nullptr;              // ... and, with spatially ordered storage, upload index -> storage slot (built on first use per geometry)
    uint64_t wire_inv_gen = 0;
    size_t wire_cap = 0;                       // pixels both hold
    // An upload in progress.  The registerUpdate()-layout arrays of ALL its entries sit in one device arena (sized at gsr_upload_begin,
    // kept between uploads: an animated sequence re-stages every cook), in upload order; gsr_upload_end orders and packs them (k_pack).
    char* stage = nullptr;
    size_t stage_cap = 0;
    char* stage_raw = nullptr;                 // ... and the raw float arrays of ONE gsr_upload_append_raw call (quantised into the arena)
    size_t stage_raw_cap = 0;
    uint32_t *up_kA = nullptr, *up_kB = nullptr, *up_vA = nullptr, *up_vB = nullptr;   // Morton codes / upload indices, ping-pong (kept between uploads)
    size_t up_sort_cap = 0;
    float* up_part = nullptr;                  // bounding-box partials
    hipEvent_t up_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double up_t_begin = 0.0, up_h2d_ms = 0.0, up_quant_ms = 0.0;
};

template <typename T>
static int dev_alloc(T** p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return GSR_OK;
}
template <typename T>
static void dev_free(T*& p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------
// for the other translation units of the library (gsr_multi.cpp)
#ifdef GSR_KPROF
extern "C" int gsr_debug_kprof_blocks(unsigned int* out8192) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out8192, HIP_SYMBOL(g_kprof_blk), sizeof(g_kprof_blk)) == hipSuccess ? 0 : -1;
}
extern "C" int gsr_debug_kprof(unsigned long long* out128) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_kprof), sizeof(g_kprof)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef SW_PROFILE
extern "C" int gsr_debug_sw_profile(unsigned long long* out8) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_sw_prof), 64);
    return 0;
}
#endif
#ifdef GSR_DEBUG_XCC
extern "C" int gsr_debug_xcc(unsigned* out4, int reset) {
    hipDeviceSynchronize();
    if (out4) hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_dbg_xcc), 16);
    if (reset) { void* p = nullptr; hipGetSymbolAddress(&p, HIP_SYMBOL(g_dbg_xcc)); hipMemset(p, 0, 16); }
    return 0;
}
#endif
#ifdef BL_PROFILE
extern "C" int gsr_debug_blend_profile(unsigned long long* out, int reset) {   // out: [BLP_MAX_WG][4][16]
    hipDeviceSynchronize();
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blend_prof), sizeof(unsigned long long) * BLP_MAX_WG * 4 * 16);
    if (reset) { void* p = nullptr; hipGetSymbolAddress(&p, HIP_SYMBOL(g_blend_prof)); hipMemset(p, 0, sizeof(unsigned long long) * BLP_MAX_WG * 4 * 16); }
    return 0;
}
#endif
__attribute__((visibility("hidden"))) int gsr_internal_set_error(int code, const char* text) { return set_err(code, "%s", text); }
void gsr_internal_comm_release(gsr_context* c);
bool gsr_internal_comm_set_user_stream(gsr_context* c, void* stream, void* own);
int gsr_internal_comm_sync(gsr_context* c);

extern "C" int gsr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* gsr_last_error(void) { return g_err; }
extern "C" const char* gsr_version(void) { return GSR_VERSION_STR; }

static int finish_open_frames(gsr_context* c);

static int sync_all(gsr_context* c)
{
    int frc = finish_open_frames(c);   // deferred frames: look at their pair counts now
    if (frc) return frc;
    for (int k = 0; k < GSR_MAX_SLOTS; ++k)
        if (c->slot[k].own) HIP_TRY(hipStreamSynchronize(c->slot[k].own));
    if (c->stream) HIP_TRY(hipStreamSynchronize(c->stream));
    return GSR_OK;
}

static bool slot_init(FrameSlot& sl)
{
    bool ok = hipStreamCreateWithFlags(&sl.own, hipStreamNonBlocking) == hipSuccess;
    sl.stream = sl.own;
    ok = ok && hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&sl.ev_user, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&sl.ev_pairs, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.counters), 8 * sizeof(unsigned long long)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.d_n), sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemset(sl.d_n, 0, sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.totals), 512 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.lazy_ctr), 2 * sizeof(unsigned long long)) == hipSuccess;
    ok = ok && hipMemset(sl.lazy_ctr, 0, 2 * sizeof(unsigned long long)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.colour_evals), 256 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemset(sl.colour_evals, 0, 256 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.hpyr), GSR_PYR_FLOATS * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.hpyr_next), GSR_PYR_FLOATS * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.hraw), (size_t)GSR_MAX_TILES_SIDE * GSR_MAX_TILES_SIDE * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.hstat), (size_t)GSR_MAX_TILES_SIDE * GSR_MAX_TILES_SIDE * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.hpyr2), GSR_PYR_FLOATS * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.slab), (GSR_SLAB_BINS + 8) * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemset(sl.slab, 0, (GSR_SLAB_BINS + 8) * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.ccnt), CC_MAX_GROUPS * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.bkt_key), (size_t)BK_BUCKETS * BK_CAP * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.bkt_val), (size_t)BK_BUCKETS * BK_CAP * sizeof(uint2)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.bkt_cnt), (size_t)BK_BUCKETS * BK_STRIDE * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemset(sl.bkt_cnt, 0, (size_t)BK_BUCKETS * BK_STRIDE * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.d_counts), 4 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemset(sl.d_counts, 0, 4 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&sl.h_end), sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    if (ok) sl.h_end[0] = 0ull;
    ok = ok && hipHostGetDevicePointer(reinterpret_cast<void**>(&sl.h_end_dev), sl.h_end, 0) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.st_scan), 512 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemset(sl.st_scan, 0, 512 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.partial), (size_t)(GSR_MAX_TILES_SIDE / 8) * (GSR_MAX_TILES_SIDE / 8) * sizeof(GsrTilePartial)) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.sup_work), 512 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemset(sl.sup_work, 0, 512 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&sl.h_total), 4 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    if (ok) { sl.h_total[0] = 0ull; sl.h_total[1] = 0ull; sl.h_total[2] = 0ull; sl.h_total[3] = 0ull; }
    ok = ok && hipHostGetDevicePointer(reinterpret_cast<void**>(&sl.h_total_dev), sl.h_total, 0) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&sl.h_counters), 8 * sizeof(unsigned long long), 0) == hipSuccess;
    if (ok) for (int j = 0; j < 8; ++j) sl.h_counters[j] = 0;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl.d_frame), 8 * sizeof(unsigned long long)) == hipSuccess;
    ok = ok && hipMemset(sl.d_frame, 0, 8 * sizeof(unsigned long long)) == hipSuccess;
    ok = ok && hipMemset(sl.counters, 0, 8 * sizeof(unsigned long long)) == hipSuccess;
    if (ok) {
        for (int k = 0; k < GSR_STAGE_EVENTS; ++k) sl.ev[k] = nullptr;
        sl.ev_ok = true;
        for (int k = 0; k < GSR_STAGE_EVENTS && ok; ++k) ok = hipEventCreate(&sl.ev[k]) == hipSuccess;
    }
    return ok;
}

static void slot_free_splat_arrays(FrameSlot& sl)
{
    dev_free(sl.rec); dev_free(sl.keyA); dev_free(sl.keyB); dev_free(sl.valA); dev_free(sl.valB); dev_free(sl.blk_cnt);
    dev_free(sl.zwin); dev_free(sl.cseg);
    sl.sort_valid = false;
}

static void slot_destroy(FrameSlot& sl)
{
    slot_free_splat_arrays(sl);
    dev_free(sl.hist); dev_free(sl.totals);
    dev_free(sl.pvA);
    dev_free(sl.sstart); dev_free(sl.send); dev_free(sl.tile_work); dev_free(sl.order); dev_free(sl.sup_work); dev_free(sl.fb);
    dev_free(sl.hpyr); dev_free(sl.hpyr_next); dev_free(sl.hraw); dev_free(sl.hstat); dev_free(sl.hpyr2); dev_free(sl.slab); dev_free(sl.tile_work_a); dev_free(sl.tbuf); dev_free(sl.ccnt); dev_free(sl.bkt_key); dev_free(sl.bkt_val); dev_free(sl.bkt_cnt); dev_free(sl.d_counts); dev_free(sl.st_scan); dev_free(sl.partial);
    if (sl.h_end) (void)hipHostFree(sl.h_end); dev_free(sl.depth_stage); dev_free(sl.dpyr); dev_free(sl.dactive);
    dev_free(sl.redo); dev_free(sl.lazy_ctr); dev_free(sl.colour_evals);
    dev_free(sl.counters); dev_free(sl.d_n);
    if (sl.h_total) (void)hipHostFree(sl.h_total);
    if (sl.h_counters) (void)hipHostFree(sl.h_counters);
    dev_free(sl.d_frame);
    if (sl.ev_ok)
        for (int k = 0; k < GSR_STAGE_EVENTS; ++k)
            if (sl.ev[k]) (void)hipEventDestroy(sl.ev[k]);
    if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    if (sl.ev_user) (void)hipEventDestroy(sl.ev_user);
    for (int k = 0; k < GSR_HOST_BANDS_MAX; ++k) if (sl.ev_band[k]) (void)hipEventDestroy(sl.ev_band[k]);
    if (sl.ev_pairs) (void)hipEventDestroy(sl.ev_pairs);
    if (sl.own) (void)hipStreamDestroy(sl.own);
}

extern "C" int gsr_create(int device, gsr_context** out)
{
    if (!out) return set_err(GSR_E_INVALID, "gsr_create: out is NULL");
    *out = nullptr;
    int ndev = gsr_device_count();
    if (ndev <= 0) return set_err(GSR_E_NO_DEVICE, "gsr_create: no HIP device visible");
    if (device < 0 || device >= ndev) return set_err(GSR_E_NO_DEVICE, "gsr_create: device %d out of range (0..%d)", device, ndev - 1);
    HIP_TRY(hipSetDevice(device));
    gsr_context* c = new (std::nothrow) gsr_context();
    if (!c) return set_err(GSR_E_OOM, "gsr_create: host allocation failed");
    c->device = device;
    if (const char* e = std::getenv("GSR_ORDER_KEEP")) c->opt_order_keep = std::atoi(e);   // (A/B hook)
    if (const char* e = std::getenv("GSR_SCATTER_DIRECT")) c->opt_scatter_direct = std::atoi(e);   // (A/B hook: -1 the general scatter, 0 never direct, 1 direct by frame size, 2 always direct)
    if (const char* e = std::getenv("GSR_K1_SCATTER")) c->opt_k1_scatter = std::atoi(e);   // (A/B hook)
    if (const char* e = std::getenv("GSR_HOST_BANDS")) c->opt_host_bands = std::min(std::max(std::atoi(e), 1), GSR_HOST_BANDS_MAX);   // (A/B hook)
    if (const char* e = std::getenv("GSR_MID_SORT")) c->opt_mid_sort = std::atoi(e);       // (A/B hook)
    if (const char* e = std::getenv("GSR_FUSE_ORDER")) c->opt_fuse_order = std::atoi(e);   // (A/B hook)
    if (const char* e = std::getenv("GSR_SLAB_FRAC")) { const int v = std::atoi(e); if (v >= 1 && v <= 255) c->slab_frac = v; }   // (A/B hook)
    if (const char* e = std::getenv("GSR_SLAB_MIN")) { const int v = std::atoi(e); if (v >= 1) c->slab_min = v; }                 // (A/B hook)
    if (const char* e = std::getenv("GSR_SLAB_MAX")) { const int v = std::atoi(e); if (v >= 1) c->slab_max = v; }                 // (A/B hook)
    if (const char* e = std::getenv("GSR_BN_ITEMS")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4) c->opt_bn_items = v; }   // (A/B hook)
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return set_err(GSR_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    c->stream = c->own_stream;
    bool ok = true;
    for (int k = 0; k < GSR_MAX_SLOTS && ok; ++k) ok = slot_init(c->slot[k]);
    if (!ok) {
        gsr_destroy(c);
        return set_err(GSR_E_HIP, "gsr_create: allocating frame slots (streams/events/mailboxes) failed");
    }
    if (hipMalloc(reinterpret_cast<void**>(&c->prefix), 256 * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->prefix_all), 256 * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->prefix_none), 256 * 4) != hipSuccess ||
        hipMemset(c->prefix, 0xff, 256 * 4) != hipSuccess || hipMemset(c->prefix_all, 0xff, 256 * 4) != hipSuccess ||
        hipMemset(c->prefix_none, 0, 256 * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->lazy_hint), 4) != hipSuccess || hipMemset(c->lazy_hint, 0, 4) != hipSuccess) {
        gsr_destroy(c);
        return set_err(GSR_E_HIP, "gsr_create: allocating the colour-prefix tables failed");
    }
    c->st.record_bytes = (int32_t)sizeof(GsrRecord);
    c->st.pair_bytes = 8;
    *out = c;
    return GSR_OK;
}

static void free_geometry(gsr_context* c)
{
    dev_free(c->geoA); dev_free(c->geoB); dev_free(c->col); dev_free(c->colrow);
    dev_free(c->perm); dev_free(c->clusA); dev_free(c->clusB); c->nclus = 0; c->h_perm.clear();
    for (int k = 0; k < GSR_MAX_SLOTS; ++k) slot_free_splat_arrays(c->slot[k]);
    c->cap = 0; c->n = 0;
}

extern "C" void gsr_destroy(gsr_context* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    gsr_internal_comm_release(c);
    free_geometry(c);
    for (int k = 0; k < GSR_MAX_SLOTS; ++k) slot_destroy(c->slot[k]);
    dev_free(c->tile_map);
    dev_free(c->wire_zbuf); dev_free(c->wire_out); dev_free(c->wire_inv); dev_free(c->stage); dev_free(c->stage_raw);
    dev_free(c->up_kA); dev_free(c->up_kB); dev_free(c->up_vA); dev_free(c->up_vB); dev_free(c->up_part);
    for (int k = 0; k < 4; ++k) if (c->up_ev[k]) (void)hipEventDestroy(c->up_ev[k]);
    dev_free(c->prefix); dev_free(c->prefix_all); dev_free(c->prefix_none); dev_free(c->lazy_hint);
    dev_free(c->pos_order); dev_free(c->blk_pre);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c;
}

extern "C" int gsr_set_stream(gsr_context* c, void* s)
{
    if (!c) return set_err(GSR_E_INVALID, "gsr_set_stream: ctx is NULL");
    // a context with a communicator (gsr_comm_init, world > 1) keeps its kernels on a stream of its own; results are ordered on `s`
    if (gsr_internal_comm_set_user_stream(c, s, c->own_stream)) return GSR_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    c->stream = s ? reinterpret_cast<hipStream_t>(s) : c->own_stream;
    for (int k = 0; k < GSR_MAX_SLOTS; ++k) c->slot[k].stream = c->slot[k].own;   // (re-bound by the next frame)
    return GSR_OK;
}

extern "C" int gsr_set_option(gsr_context* c, int option, int value)
{
    if (!c) return set_err(GSR_E_INVALID, "gsr_set_option: ctx is NULL");
    switch (option) {
    case GSR_OPT_XCD_SWIZZLE: c->opt_swizzle = value < 0 ? 0 : (value > 3 ? 3 : value); break;
    case GSR_OPT_STAGE_TIMING: c->opt_timing = value < 0 ? 0 : (value > 2 ? 2 : value); break;
    case GSR_OPT_SORT_CACHE:
        if (value < 0 || value > 2) return set_err(GSR_E_INVALID, "gsr_set_option: sort cache is 0, 1 or 2");
        c->opt_sort_cache = value;
        c->pos_valid = false;
        for (int k = 0; k < GSR_MAX_SLOTS; ++k) c->slot[k].sort_valid = false;
        break;
    case GSR_OPT_DEBUG_FLAGS: c->opt_flags = value; break;
    case GSR_OPT_DEFERRED_CHECK: c->opt_deferred = value ? 1 : 0; break;
    case GSR_OPT_TIMING_EVERY: c->opt_timing_every = value < 1 ? 1 : (value > 1024 ? 1024 : value); break;
    case GSR_OPT_OCCLUSION_CULL: c->opt_cull = value < 0 ? 0 : (value > 3 ? 3 : value); break;
    case GSR_OPT_LAZY_COLOUR: c->opt_lazy = value < 0 ? 0 : (value > 2 ? 2 : value); break;
    case GSR_OPT_SHARD_LAYOUT: c->shard_layout = value ? 1 : 0; break;
    case GSR_OPT_CLUSTER_CULL: c->opt_cluster = value ? 1 : 0; break;
    case GSR_OPT_LOCAL_SORT: c->opt_local_sort = value < 0 ? 0 : (value > 2 ? 2 : value); break;
    case GSR_OPT_FRONT_SLAB: c->opt_slab = value < 0 ? 0 : (value > 2 ? 2 : value); break;
    case GSR_OPT_CULL_DILATE: c->cull_pol.set_dilate_option(value); break;
    case GSR_OPT_STORAGE_ORDER: c->opt_morton = value ? 1 : 0; break;   // (takes effect at the next upload)
    case GSR_OPT_SUPER_TILE:
        if (value != 0 && (value < 1 || value > 16 || (value & (value - 1))))
            return set_err(GSR_E_INVALID, "gsr_set_option: super-tile edge must be 0 (auto) or 1,2,4,8,16");
        c->opt_super = value;
        break;
    case GSR_OPT_FRAMES_IN_FLIGHT: {
        if (value < 1 || value > GSR_MAX_SLOTS)
            return set_err(GSR_E_INVALID, "gsr_set_option: frames in flight must be 1..%d", GSR_MAX_SLOTS);
        HIP_TRY(hipSetDevice(c->device));
        int rc = sync_all(c);
        if (rc) return rc;
        c->nslots = value;
        break;
    }
    default: return set_err(GSR_E_INVALID, "gsr_set_option: unknown option %d", option);
    }
    return GSR_OK;
}

// the arena's arrays for `total` splats (256-byte aligned sections)
struct UpArena {
    float *P, *alpha;
    uint16_t *Cd, *scale, *orient, *shx, *shy, *shz;
    size_t bytes;
};
static UpArena up_arena(char* base, size_t total, bool has_sh)
{
    const size_t al = 256;
    auto pad = [&](size_t b) { return (b + al - 1) / al * al; };
    UpArena a{};
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += pad(bytes); return p; };
    a.P = reinterpret_cast<float*>(take(total * 12)); a.alpha = reinterpret_cast<float*>(take(total * 4));
    a.Cd = reinterpret_cast<uint16_t*>(take(total * 6)); a.scale = reinterpret_cast<uint16_t*>(take(total * 6));
    a.orient = reinterpret_cast<uint16_t*>(take(total * 8));
    if (has_sh) {
        a.shx = reinterpret_cast<uint16_t*>(take(total * 32)); a.shy = reinterpret_cast<uint16_t*>(take(total * 32));
        a.shz = reinterpret_cast<uint16_t*>(take(total * 32));
    }
    a.bytes = off;
    return a;
}
static double up_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---------------------------------------------------------------------------
// geometry staging
extern "C" int gsr_upload_begin(gsr_context* c, int64_t total, int has_sh, const float origin[3])
{
    if (!c) return set_err(GSR_E_INVALID, "gsr_upload_begin: ctx is NULL");
    if (total < 0 || total > 0x7fffffffll) return set_err(GSR_E_INVALID, "gsr_upload_begin: bad splat count %lld", (long long)total);
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    const uint32_t n = (uint32_t)total;
    const int chunks = has_sh ? 6 : 1;
    if (n > c->cap || chunks != c->col_chunks) {
        free_geometry(c);
        const size_t cap = n ? n : 1;
        if ((rc = dev_alloc(&c->geoA, cap)) || (rc = dev_alloc(&c->geoB, cap)) || (rc = dev_alloc(&c->col, cap * chunks)) ||
            (has_sh && (rc = dev_alloc(&c->colrow, cap * 8)))) {
            free_geometry(c);
            return rc;
        }
        for (int k = 0; k < GSR_MAX_SLOTS; ++k) {
            FrameSlot& sl = c->slot[k];
            // (+256: K1 writes what it keeps at the head of its workgroup's 256 slots)
            if ((rc = dev_alloc(&sl.rec, cap)) || (rc = dev_alloc(&sl.keyA, cap + 256)) || (rc = dev_alloc(&sl.keyB, cap + 256)) ||
                (rc = dev_alloc(&sl.valA, cap + 256)) || (rc = dev_alloc(&sl.valB, cap + 256)) ||
                (rc = dev_alloc(&sl.zwin, cap)) || (rc = dev_alloc(&sl.blk_cnt, cap / 256 + 16)) ||
                (rc = dev_alloc(&sl.cseg, cap / GSR_CLUSTER + (size_t)CC_THREADS * 64 + 16))) {
                free_geometry(c);
                return rc;
            }
        }
        c->cap = (uint32_t)cap;
        c->col_chunks = chunks;
    }
    // the staging arena of the whole upload (kept between uploads; only ever grown)
    {
        const UpArena A = up_arena(nullptr, n, has_sh != 0);
        if (A.bytes > c->stage_cap) {
            dev_free(c->stage);
            c->stage_cap = 0;
            if ((rc = dev_alloc(&c->stage, A.bytes + 256))) return rc;
            c->stage_cap = A.bytes + 256;
        }
    }
    c->up_t_begin = up_now_ms(); c->up_h2d_ms = 0.0; c->up_quant_ms = 0.0;
    c->n = 0;
    c->has_sh = has_sh != 0;
    c->up_total = n;
    c->up_filled = 0;
    c->uploading = true;
    for (int k = 0; k < 3; ++k) c->origin[k] = origin ? origin[k] : 0.0f;
    for (int k = 0; k < GSR_MAX_SLOTS; ++k) c->slot[k].sort_valid = false;
    dev_free(c->perm); dev_free(c->clusA); dev_free(c->clusB);   // (the arrays are filled in upload order again)
    c->nclus = 0; c->h_perm.clear();
    return GSR_OK;
}

extern "C" int gsr_upload_append(gsr_context* c, int64_t n64, const float* P, const uint16_t* Cd, const float* alpha,
                                 const uint16_t* scale, const uint16_t* orient, const uint16_t* shx,
                                 const uint16_t* shy, const uint16_t* shz)
{
    if (!c || !c->uploading) return set_err(GSR_E_INVALID, "gsr_upload_append: no upload in progress");
    if (n64 < 0 || (uint64_t)n64 + c->up_filled > c->up_total)
        return set_err(GSR_E_INVALID, "gsr_upload_append: %lld splats exceed the %u announced", (long long)n64, c->up_total);
    if (n64 == 0) return GSR_OK;
    if (!P || !Cd || !alpha || !scale || !orient) return set_err(GSR_E_INVALID, "gsr_upload_append: NULL attribute array");
    if (c->has_sh && (!shx || !shy || !shz)) return set_err(GSR_E_INVALID, "gsr_upload_append: SH announced but arrays are NULL");
    HIP_TRY(hipSetDevice(c->device));
    const double t0 = up_now_ms();
    const uint32_t n = (uint32_t)n64;
    const size_t at = c->up_filled;
    hipStream_t us = c->slot[0].own;
    const UpArena A = up_arena(c->stage, c->up_total, c->has_sh);
    // Host -> device, nothing else: the entry lands behind the ones before it, in upload order.  (Pageable memory: the runtime stages
    // it through its own pinned buffers at the link's rate -- 56 GB/s measured on this box, the same as from pinned memory:
    // tools/ubench_h2d.hip.  Ordering, packing and the cluster bounds happen once, over the whole upload, in gsr_upload_end.)
    hipError_t e = hipSuccess;
    auto h2d = [&](void* d, const void* h, size_t bytes) { if (e == hipSuccess) e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, us); };
    h2d(A.P + 3 * at, P, (size_t)n * 12); h2d(A.alpha + at, alpha, (size_t)n * 4); h2d(A.Cd + 3 * at, Cd, (size_t)n * 6);
    h2d(A.scale + 3 * at, scale, (size_t)n * 6); h2d(A.orient + 4 * at, orient, (size_t)n * 8);
    if (c->has_sh) { h2d(A.shx + 16 * at, shx, (size_t)n * 32); h2d(A.shy + 16 * at, shy, (size_t)n * 32); h2d(A.shz + 16 * at, shz, (size_t)n * 32); }
    if (e == hipSuccess) e = hipStreamSynchronize(us);   // the caller's arrays may be freed on return
    if (e != hipSuccess) return set_err(GSR_E_HIP, "gsr_upload_append: %s", hipGetErrorString(e));
    c->up_filled += n;
    c->up_h2d_ms += up_now_ms() - t0;
    return GSR_OK;
}

static int order_and_pack(gsr_context* c);

// an upload that failed after the arrays were filled: the context goes back to "nothing uploaded"
static void drop_geometry(gsr_context* c)
{
    c->n = 0; c->nclus = 0; c->up_total = c->up_filled = 0; c->st.n_splats = 0;
    dev_free(c->perm); dev_free(c->clusA); dev_free(c->clusB); c->h_perm.clear();
    c->has_geometry = false;           // gsr_render: GSR_E_NO_GEOMETRY (the generation is NOT rewound: stamps of the lost cloud stay stale)
    c->geo_gen++;
    dev_free(c->wire_inv); c->wire_inv_gen = 0;
    c->pos_valid = false; c->prefix_valid = false;
    for (int k = 0; k < GSR_MAX_SLOTS; ++k) {
        FrameSlot& sl = c->slot[k];
        sl.sort_valid = false; sl.horizon_valid = false; sl.order_valid = false;
        sl.surv_hint = 0; sl.kept_hint = 0; sl.kept_lo = sl.kept_hi = 0;
    }
}

extern "C" int gsr_upload_append_raw(gsr_context* c, int64_t n64, const gsr_raw_attrs* a)
{
    if (!c || !c->uploading) return set_err(GSR_E_INVALID, "gsr_upload_append_raw: no upload in progress");
    if (!a) return set_err(GSR_E_INVALID, "gsr_upload_append_raw: attrs is NULL");
    if (n64 < 0 || (uint64_t)n64 + c->up_filled > c->up_total)
        return set_err(GSR_E_INVALID, "gsr_upload_append_raw: %lld splats exceed the %u announced", (long long)n64, c->up_total);
    if (n64 == 0) return GSR_OK;
    if (!a->P) return set_err(GSR_E_INVALID, "gsr_upload_append_raw: P is NULL");
    if (a->sh_scheme < 0 || a->sh_scheme > 3 || (a->sh_scheme == 1 && (!a->sh_array || a->sh_vec3_per_point < 1)) || (a->sh_scheme >= 2 && !a->sh_ptr))
        return set_err(GSR_E_INVALID, "gsr_upload_append_raw: bad spherical-harmonics description");
    if (c->has_sh != (a->sh_scheme != 0)) return set_err(GSR_E_INVALID, "gsr_upload_append_raw: SH presence differs from gsr_upload_begin");
    HIP_TRY(hipSetDevice(c->device));
    const double t0 = up_now_ms();
    const uint32_t n = (uint32_t)n64;
    const size_t at = c->up_filled;
    hipStream_t us = c->slot[0].own;
    const UpArena A = up_arena(c->stage, c->up_total, c->has_sh);
    const size_t al = 256;
    auto pad = [&](size_t b) { return (b + al - 1) / al * al; };
    // the raw float arrays of THIS entry: their own scratch (kept between calls), quantised into the arena behind the entries before it
    const int nsh = a->sh_scheme == 2 ? 15 : (a->sh_scheme == 3 ? 45 : 0);
    int sh_live = 0;                 // schemes 2 / 3: the arrays before the first gap
    if (nsh) while (sh_live < nsh && a->sh_ptr[sh_live]) ++sh_live;
    const size_t rC = a->Cd ? pad((size_t)n * 12) : 0, rS = a->scale ? pad((size_t)n * 12) : 0, rO = a->orient ? pad((size_t)n * 16) : 0;
    const size_t rArr = a->sh_scheme == 1 ? pad((size_t)n * std::min(a->sh_vec3_per_point, 16) * 12) : 0;
    const size_t rEach = a->sh_scheme == 2 ? pad((size_t)n * 12) : (a->sh_scheme == 3 ? pad((size_t)n * 4) : 0);
    const size_t need = rC + rS + rO + rArr + rEach * sh_live + al;
    if (need > c->stage_raw_cap) {
        HIP_TRY(hipStreamSynchronize(us));
        dev_free(c->stage_raw);
        c->stage_raw_cap = 0;
        int rc = dev_alloc(&c->stage_raw, need);
        if (rc) return rc;
        c->stage_raw_cap = need;
    }
    char* base = c->stage_raw;
    auto take = [&](size_t bytes) { char* p = base; base += bytes; return p; };
    hipError_t e = hipSuccess;
    auto h2d = [&](void* d, const void* h, size_t bytes) { if (e == hipSuccess) e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, us); };
    h2d(A.P + 3 * at, a->P, (size_t)n * 12);
    std::vector<float> ones;
    if (a->alpha) h2d(A.alpha + at, a->alpha, (size_t)n * 4);
    else { ones.assign(n, 1.0f); h2d(A.alpha + at, ones.data(), (size_t)n * 4); }     // missing opacity: opaque
    const float *rCd = nullptr, *rSc = nullptr, *rOr = nullptr;
    if (a->Cd) { float* p = reinterpret_cast<float*>(take(rC)); h2d(p, a->Cd, (size_t)n * 12); rCd = p; }
    if (a->scale) { float* p = reinterpret_cast<float*>(take(rS)); h2d(p, a->scale, (size_t)n * 12); rSc = p; }
    if (a->orient) { float* p = reinterpret_cast<float*>(take(rO)); h2d(p, a->orient, (size_t)n * 16); rOr = p; }
    GsrRawSh sh{};
    sh.scheme = a->sh_scheme;
    sh.vec3_per_point = a->sh_vec3_per_point > 16 ? 16 : a->sh_vec3_per_point;
    if (a->sh_scheme == 1) {
        // (only the first 16 vec3 of a longer array are used: copy them compactly)
        float* p = reinterpret_cast<float*>(take(rArr));
        if (a->sh_vec3_per_point <= 16) h2d(p, a->sh_array, (size_t)n * a->sh_vec3_per_point * 12);
        else if (e == hipSuccess)
            e = hipMemcpy2DAsync(p, (size_t)16 * 12, a->sh_array, (size_t)a->sh_vec3_per_point * 12, (size_t)16 * 12, n, hipMemcpyHostToDevice, us);
        sh.array = p;
    } else {
        for (int j = 0; j < sh_live; ++j) { float* p = reinterpret_cast<float*>(take(rEach)); h2d(p, a->sh_ptr[j], a->sh_scheme == 2 ? (size_t)n * 12 : (size_t)n * 4); sh.ptr[j] = p; }
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_quantize_raw, dim3(div_up(n, 256)), dim3(256), 0, us, n, rCd, rSc, rOr, sh, A.Cd + 3 * at, A.scale + 3 * at, A.orient + 4 * at,
                           c->has_sh ? A.shx + 16 * at : (uint16_t*)nullptr, c->has_sh ? A.shy + 16 * at : (uint16_t*)nullptr, c->has_sh ? A.shz + 16 * at : (uint16_t*)nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(us);   // the caller's arrays may be freed on return
    if (e != hipSuccess) return set_err(GSR_E_HIP, "gsr_upload_append_raw: %s", hipGetErrorString(e));
    c->up_filled += n;
    c->up_h2d_ms += up_now_ms() - t0;
    return GSR_OK;
}

extern "C" int gsr_upload_end(gsr_context* c)
{
    if (!c || !c->uploading) return set_err(GSR_E_INVALID, "gsr_upload_end: no upload in progress");
    c->uploading = false;
    if (c->up_filled != c->up_total)
        return set_err(GSR_E_INVALID, "gsr_upload_end: %u of %u announced splats were appended", c->up_filled, c->up_total);
    c->n = c->up_total;
    c->bbox_ok = false;
    if (c->n > 0) {
        HIP_TRY(hipSetDevice(c->device));
        int rc = order_and_pack(c);
        if (rc) {
            // nothing half-made may stay renderable: no geometry, no cached order, no horizons, no hints (a later gsr_render says
            // GSR_E_NO_GEOMETRY until a complete upload has succeeded)
            drop_geometry(c);
            return rc;
        }
    }
    c->geo_gen++;
    c->has_geometry = true;
    for (int k = 0; k < GSR_MAX_SLOTS; ++k) c->slot[k].sort_valid = false;
    c->prefix_valid = false;           // lazy colour: the first frame of a new cloud colours every list completely
    c->order_pays = false;
    c->cull_pol.on_upload(); c->slab_pol.on_upload();
    for (int k = 0; k < GSR_MAX_SLOTS; ++k) { c->slot[k].surv_hint = 0; c->slot[k].kept_hint = 0; c->slot[k].kept_lo = c->slot[k].kept_hi = 0; c->slot[k].horizon_valid = false; c->slot[k].local_pol.on_upload(); c->slot[k].slab_kept1 = c->slot[k].slab_kept2 = 0; }
    c->lazy_pays = false;              // ... and in automatic mode the first frames are eager until the kernels say it pays
    // (the kernels' verdicts on the last frame of the PREVIOUS cloud must not reach the first frame of this one through the device word)
    if (c->lazy_hint && hipMemsetAsync(c->lazy_hint, 0, 4, c->slot[0].own) != hipSuccess) return set_err(GSR_E_HIP, "upload: could not reset the policy word");
    if (c->lazy_hint && hipStreamSynchronize(c->slot[0].own) != hipSuccess) return set_err(GSR_E_HIP, "upload: could not reset the policy word");
    c->st.n_splats = c->n;
    c->st.uploads += 1;
    c->st.upload_ms[0] = c->up_h2d_ms;
    c->st.upload_ms[3] = up_now_ms() - c->up_t_begin;
    return GSR_OK;
}

extern "C" int gsr_upload_abort(gsr_context* c)
{
    if (!c) return set_err(GSR_E_INVALID, "gsr_upload_abort: ctx is NULL");
    // an upload that failed half-way leaves no geometry behind: rendering needs a complete new upload
    if (c->uploading) { c->uploading = false; c->n = 0; c->up_total = c->up_filled = 0; c->st.n_splats = 0; }
    return GSR_OK;
}

extern "C" int gsr_upload(gsr_context* c, int64_t n, const float* P, const uint16_t* Cd, const float* alpha,
                          const uint16_t* scale, const uint16_t* orient, const uint16_t* shx, const uint16_t* shy,
                          const uint16_t* shz, const float origin[3])
{
    const int has_sh = (shx && shy && shz) ? 1 : 0;
    int rc = gsr_upload_begin(c, n, has_sh, origin);
    if (rc) return rc;
    rc = gsr_upload_append(c, n, P, Cd, alpha, scale, orient, shx, shy, shz);
    if (!rc) rc = gsr_upload_end(c);
    if (rc) (void)gsr_upload_abort(c);
    return rc;
}

// ---------------------------------------------------------------------------
// multi-GPU shard helpers
extern "C" int gsr_set_row_shard(gsr_context* c, int index, int count)
{
    if (!c) return set_err(GSR_E_INVALID, "gsr_set_row_shard: ctx is NULL");
    if (count < 1 || index < 0 || index >= count) return set_err(GSR_E_INVALID, "gsr_set_row_shard: bad shard %d/%d", index, count);
    c->shard_index = index;
    c->shard_count = count;
    return GSR_OK;
}

extern "C" int gsr_band_rows(int height, int index, int count)
{
    (void)index;  // every shard uses the same (padded) band height so that gathers are uniform
    if (height <= 0 || count < 1) return 0;
    const int tiles_y = (height + GSR_TILE - 1) / GSR_TILE;
    return ((tiles_y + count - 1) / count) * GSR_TILE;
}

__attribute__((visibility("hidden"))) int gsr_internal_shard_layout(gsr_context* c) { return c ? c->shard_layout : 0; }
__attribute__((visibility("hidden"))) int gsr_internal_stitch(gsr_context* c, const float* gathered, int count, int width, int height, float* out, void* stream);
extern "C" int gsr_stitch_bands(gsr_context* c, const float* gathered, int count, int width, int height, float* out)
{
    return gsr_internal_stitch(c, gathered, count, width, height, out, c ? (void*)c->stream : nullptr);
}
// (the stream: the multi-GPU gather stitches on its transfer stream, gsr_multi.cpp)
__attribute__((visibility("hidden"))) int gsr_internal_stitch(gsr_context* c, const float* gathered, int count, int width, int height, float* out, void* stream)
{
    // (the bands were rendered with this context's shard layout: GSR_OPT_SHARD_LAYOUT)
    if (!c || !gathered || !out || count < 1 || width <= 0 || height <= 0)
        return set_err(GSR_E_INVALID, "gsr_stitch_bands: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t npx = (size_t)width * height;
    hipLaunchKernelGGL(k_stitch_bands, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(gathered), count,
                       (c->shard_layout == 1 && count > 1) ? ((height + GSR_TILE - 1) / GSR_TILE + count - 1) / count : 0,
                       gsr_band_rows(height, 0, count), width, height, reinterpret_cast<float4*>(out));
    HIP_TRY(hipGetLastError());
    return GSR_OK;
}

// ---------------------------------------------------------------------------
// scan + sort drivers (all on the slot's stream)
static int ensure_u32(uint32_t** p, size_t* cap, size_t need)
{
    if (need <= *cap) return GSR_OK;
    dev_free(*p);
    *cap = 0;
    size_t want = need + need / 4 + 1024;
    int rc = dev_alloc(p, want);
    if (rc) return rc;
    *cap = want;
    return GSR_OK;
}

template <typename V, int DBITS, bool GATHER, int ITEMS>
static int radix_pass(FrameSlot& sl, uint32_t* kA, V* vA, uint32_t* kB, V* vB, uint32_t n, const uint32_t* n_dev,
                      int shift, uint32_t nblk, bool contig, uint32_t* n_out = nullptr, const uint32_t* src_cnt = nullptr, uint32_t lo = 0u)
{
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_radix_hist<DBITS, GATHER, false, ITEMS>), dim3(nblk), dim3(RS_THREADS), 0, sl.stream, kA, n,
                       n_dev, shift, sl.hist, nblk, contig, src_cnt, lo);
    hipLaunchKernelGGL(k_scan_rows, dim3(1u << DBITS), dim3(SC_THREADS), 0, sl.stream, sl.hist, nblk, sl.totals, n_dev, n,
                       (uint32_t)RS_THREADS * (uint32_t)ITEMS);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_radix_scatter<V, DBITS, GATHER, false, ITEMS>), dim3(nblk), dim3(RS_THREADS), 0, sl.stream, kA,
                       vA, kB, vB, n, n_dev, shift, sl.hist, sl.totals, nblk, contig, n_out, src_cnt, lo);
    HIP_TRY(hipGetLastError());
    return GSR_OK;
}

// stable LSD sort on key bits [0, bits); ping-pongs (kA,vA) <-> (kB,vB) and leaves the result
// in (kA,vA) by swapping the pointers.  9-bit digits are used when they save a pass.
// compact_to != NULL: the input is K1's block-compacted layout (src_cnt items at the head of every 256 slots, *src_n_dev <= n slots);
// the first pass gathers them and writes their number to *compact_to (device); the remaining passes and the caller's
// later kernels read it there.
template <typename V>
static int radix_sort(FrameSlot& sl, uint32_t*& kA, V*& vA, uint32_t*& kB, V*& vB, uint32_t n, int bits,
                      bool allow9 = true, uint32_t* compact_to = nullptr, bool contig = false, const uint32_t* src_cnt = nullptr,
                      const uint32_t* src_n_dev = nullptr /* compacting pass: the number of source SLOTS, on the device (<= n) */,
                      bool mid = false /* a frame expected to keep <= ~1.5 M keys: RS_ITEMS_MID keys per thread (shorter workgroups, more of them) */)
{
    if (n == 0) {
        if (compact_to) HIP_TRY(hipMemsetAsync(compact_to, 0, 4, sl.stream));
        return GSR_OK;
    }
    if (bits <= 0) bits = 1;
    const uint32_t nblk = div_up(n, (uint32_t)RS_THREADS * (uint32_t)(mid ? RS_ITEMS_MID : RS_ITEMS));
    int rc = ensure_u32(&sl.hist, &sl.hist_cap, (size_t)512 * nblk + 8);
    if (rc) return rc;
    const int p8 = (bits + 7) / 8, p9 = (bits + 8) / 9;
    const bool use9 = allow9 && p9 < p8;
    const int passes = use9 ? p9 : p8, width = use9 ? 9 : 8;
    for (int p = 0; p < passes; ++p) {
        const bool skip = compact_to && p == 0;
        const uint32_t* n_dev = compact_to ? (p > 0 ? compact_to : src_n_dev) : nullptr;
#define GSR_PASS(W, I) (skip ? radix_pass<V, W, true, I>(sl, kA, vA, kB, vB, n, n_dev, p * width, nblk, contig, compact_to, src_cnt) \
                             : radix_pass<V, W, false, I>(sl, kA, vA, kB, vB, n, n_dev, p * width, nblk, contig))
        if (use9) rc = mid ? GSR_PASS(9, RS_ITEMS_MID) : GSR_PASS(9, RS_ITEMS);
        else rc = mid ? GSR_PASS(8, RS_ITEMS_MID) : GSR_PASS(8, RS_ITEMS);
#undef GSR_PASS
        if (rc) return rc;
        uint32_t* t = kA; kA = kB; kB = t;
        V* tv = vA; vA = vB; vB = tv;
    }
    return GSR_OK;
}

// ---------------------------------------------------------------------------
// End of an upload: the splats go into MORTON ORDER of their positions (k_cluster.h) -- a stable sort of 30-bit codes, so
// equal codes keep their upload order and the storage order is a pure function of the positions -- and every 64 consecutive
// slots get their cluster bounds.  c->perm[j] = upload index of the splat in slot j (NULL: upload order kept).
static void cluster_grid(uint32_t nclus, int* rounds, uint32_t* ngroups)
{
    int r = 1;
    while ((uint64_t)CC_THREADS * (uint64_t)r * (uint64_t)CC_MAX_GROUPS < (uint64_t)nclus) ++r;
    *rounds = r;
    *ngroups = nclus ? div_up(nclus, (uint32_t)CC_THREADS * (uint32_t)r) : 0u;
}

static int order_and_pack(gsr_context* c)
{
    const uint32_t n = c->n;
    FrameSlot& sl = c->slot[0];
    hipStream_t us = sl.stream;
    int rc = GSR_OK;
    const UpArena A = up_arena(c->stage, c->up_total, c->has_sh);
    for (int k = 0; k < 4; ++k)
        if (!c->up_ev[k]) HIP_TRY(hipEventCreate(&c->up_ev[k]));
    HIP_TRY(hipEventRecord(c->up_ev[0], us));
    // bounding box of the positions: bounds distance^2 to any camera, i.e. the sort-key range per frame -- and the Morton grid
    {
        const int grid = 512;
        if (!c->up_part && (rc = dev_alloc(&c->up_part, (size_t)grid * 6))) return rc;
        hipLaunchKernelGGL(k_bbox_partials, dim3(grid), dim3(256), 0, us, A.P, n, c->up_part);
        std::vector<float> hp((size_t)grid * 6);
        hipError_t e = hipMemcpyAsync(hp.data(), c->up_part, hp.size() * 4, hipMemcpyDeviceToHost, us);
        if (e == hipSuccess) e = hipStreamSynchronize(us);
        if (e != hipSuccess) return set_err(GSR_E_HIP, "gsr_upload_end: %s", hipGetErrorString(e));
        bool ok = true;
        for (int k = 0; k < 3; ++k) { c->bb_lo[k] = 3.0e38; c->bb_hi[k] = -3.0e38; }
        for (int b = 0; b < grid; ++b)
            for (int k = 0; k < 3; ++k) {
                const float lo = hp[(size_t)b * 6 + k], hi = hp[(size_t)b * 6 + 3 + k];
                ok = ok && std::isfinite(lo) && std::isfinite(hi);
                c->bb_lo[k] = std::min(c->bb_lo[k], (double)lo);
                c->bb_hi[k] = std::max(c->bb_hi[k], (double)hi);
            }
        c->bbox_ok = ok;
    }
    // the storage order: perm[j] = upload index of the splat in slot j (NULL: upload order)
    dev_free(c->perm);
    if (c->opt_morton && c->bbox_ok && n > 1) {
        bool have = true;
        if ((size_t)n > c->up_sort_cap) {
            dev_free(c->up_kA); dev_free(c->up_kB); dev_free(c->up_vA); dev_free(c->up_vB);
            c->up_sort_cap = 0;
            const size_t want = (size_t)n + n / 8 + 1024;
            have = !(dev_alloc(&c->up_kA, want) || dev_alloc(&c->up_kB, want) || dev_alloc(&c->up_vA, want) || dev_alloc(&c->up_vB, want));
            if (have) c->up_sort_cap = want;
            else {
                // without the scratch the splats simply stay in upload order (perm = NULL: ties then break by upload index, the
                // documented meaning of an unordered store) and get their clusters
                dev_free(c->up_kA); dev_free(c->up_kB); dev_free(c->up_vA); dev_free(c->up_vB);
                (void)hipGetLastError();
            }
        }
        if (have && !(rc = dev_alloc(&c->perm, (size_t)n))) {
            float lo[3], sc[3];
            for (int k = 0; k < 3; ++k) {
                lo[k] = (float)c->bb_lo[k];
                const double ext = c->bb_hi[k] - c->bb_lo[k];
                sc[k] = ext > 0.0 ? (float)(1023.999 / ext) : 0.0f;
                if (!std::isfinite(sc[k])) sc[k] = 0.0f;
            }
            uint32_t *kA = c->up_kA, *kB = c->up_kB, *vA = c->up_vA, *vB = c->up_vB;
            hipLaunchKernelGGL(k_morton_codes, dim3(div_up(n, 256)), dim3(256), 0, us, A.P, n, lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], kA, vA);
            rc = radix_sort(sl, kA, vA, kB, vB, n, 30, true, (uint32_t*)nullptr, RS_XCD_DEPTH != 0);   // (leaves the result in what kA / vA name now)
            if (!rc && hipMemcpyAsync(c->perm, vA, (size_t)n * 4, hipMemcpyDeviceToDevice, us) != hipSuccess) rc = set_err(GSR_E_HIP, "gsr_upload_end: storage order");
            if (rc) { dev_free(c->perm); return rc; }
        } else if (have) {
            (void)hipGetLastError();
            rc = GSR_OK;           // (no room for the permutation: upload order)
        }
    }
    HIP_TRY(hipEventRecord(c->up_ev[1], us));
    c->nclus = div_up(n, GSR_CLUSTER);
    dev_free(c->clusA); dev_free(c->clusB);
    if ((rc = dev_alloc(&c->clusA, c->nclus)) || (rc = dev_alloc(&c->clusB, c->nclus))) return rc;
    const GsrPackSrc src{A.P, A.alpha, A.Cd, A.scale, A.orient, A.shx, A.shy, A.shz};
    if (c->has_sh)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pack<true>), dim3(c->nclus), dim3(GSR_PACK_THREADS), 0, us, n, c->cap, src, c->perm, c->geoA, c->geoB, c->col, c->colrow, c->clusA, c->clusB);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pack<false>), dim3(c->nclus), dim3(GSR_PACK_THREADS), 0, us, n, c->cap, src, c->perm, c->geoA, c->geoB, c->col, c->colrow, c->clusA, c->clusB);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(c->up_ev[2], us);
    if (e == hipSuccess) e = hipStreamSynchronize(us);
    if (e != hipSuccess) return set_err(GSR_E_HIP, "gsr_upload_end: packing the splats: %s", hipGetErrorString(e));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->up_ev[0], c->up_ev[1]) == hipSuccess) c->st.upload_ms[1] = ms;
    if (hipEventElapsedTime(&ms, c->up_ev[1], c->up_ev[2]) == hipSuccess) c->st.upload_ms[2] = ms;
    return GSR_OK;
}

// ---------------------------------------------------------------------------
static inline float M4h(const float* m, int r, int c) { return m[c * 4 + r]; }

// upper bound of the largest singular value (squared) of the 3x3 block of a GL column-major 4x4: power iteration on M^T M
// in double, padded; the Frobenius norm (always an upper bound) if the iteration misbehaves
static double sigma_max_sq(const float* m)
{
    double a[3][3], g[3][3];
    double fro = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) { a[r][k] = M4h(m, r, k); fro += a[r][k] * a[r][k]; }
    if (!(fro > 0.0) || !std::isfinite(fro)) return fro;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) g[i][k] = a[0][i] * a[0][k] + a[1][i] * a[1][k] + a[2][i] * a[2][k];
    double v[3] = {1.0, 0.7, 0.4}, lam = 0.0;
    for (int it = 0; it < 64; ++it) {
        double w[3];
        for (int i = 0; i < 3; ++i) w[i] = g[i][0] * v[0] + g[i][1] * v[1] + g[i][2] * v[2];
        const double nrm = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        if (!(nrm > 0.0)) return fro;
        lam = nrm / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int i = 0; i < 3; ++i) v[i] = w[i] / nrm;
    }
    // the iteration converges from below: pad generously (it only feeds a conservative cull), never above Frobenius
    const double padded = lam * 1.02 + 1e-12;
    return (std::isfinite(padded) && padded < fro) ? padded : fro;
}

// Depth-tested frames: the range of window depths the list entries' nine depth bits spread over (gsr_device.h: gsr_zq) -- that of the
// cloud's bounding box as this camera sees it.  Nothing depends on the range being right: the code clamps, and it is monotone whatever
// the constants are; a bad range only filters less.  Clouds of 2^23 splats or more have no spare index bits: no codes.
static void frame_depth_codes(const gsr_context* c, const gsr_camera* cam, GsrFrame* f)
{
    f->idx_mask = 0xffffffffu; f->zq0 = 0.0f; f->zqs = 0.0f;
    // (nor frames whose depth buffer is expected to be clear -- the plain kernel behind its guard -- where the codes would only cost)
    if (c->n >= (1u << GSR_ZQ_SHIFT) || !c->bbox_ok || !c->depth_active || (c->opt_flags & GSR_FLAG_NO_ZCODES)) return;
    double zlo = 1.0, zhi = 0.0;
    bool behind = false;
    for (int k = 0; k < 8; ++k) {
        double p[4], e[4], q[4];
        for (int a = 0; a < 3; ++a) p[a] = ((k >> a) & 1) ? c->bb_hi[a] : c->bb_lo[a];
        p[3] = 1.0;
        for (int r = 0; r < 4; ++r) e[r] = M4h(cam->obj_view, r, 0) * p[0] + M4h(cam->obj_view, r, 1) * p[1] + M4h(cam->obj_view, r, 2) * p[2] + M4h(cam->obj_view, r, 3);
        for (int r = 0; r < 4; ++r) q[r] = M4h(cam->proj, r, 0) * e[0] + M4h(cam->proj, r, 1) * e[1] + M4h(cam->proj, r, 2) * e[2] + M4h(cam->proj, r, 3) * e[3];
        if (!(q[3] > 1e-9)) { behind = true; continue; }
        const double zw = 0.5 * (q[2] / q[3]) + 0.5;
        if (std::isfinite(zw)) { zlo = std::min(zlo, zw); zhi = std::max(zhi, zw); }
    }
    if (behind) zlo = 0.0;
    zlo = std::min(std::max(zlo, 0.0), 1.0); zhi = std::min(std::max(zhi, 0.0), 1.0);
    if (!(zhi > zlo)) return;
    f->idx_mask = (1u << GSR_ZQ_SHIFT) - 1u;
    f->zq0 = (float)zlo;
    f->zqs = (float)(512.0 / (zhi - zlo));
    if (!std::isfinite(f->zqs)) { f->idx_mask = 0xffffffffu; f->zqs = 0.0f; }
}

static void build_frame(const gsr_context* c, const gsr_camera* cam, GsrFrame* f)
{
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 4; ++k) {
            f->ov[r * 4 + k] = M4h(cam->obj_view, r, k);
            f->vw[r * 4 + k] = M4h(cam->view, r, k);
        }
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 4; ++k) f->pr[r * 4 + k] = M4h(cam->proj, r, k);
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) {
            f->ob[r * 3 + k] = M4h(cam->object, r, k);
            f->io[r * 3 + k] = M4h(cam->inv_object, r, k);
        }
    for (int k = 0; k < 3; ++k) { f->cam[k] = cam->cam_pos[k]; f->origin[k] = c->origin[k]; }
    f->idx_mask = 0xffffffffu; f->zq0 = 0.0f; f->zqs = 0.0f;      // (depth-tested frames: frame_depth_codes)
    // CalcCovariance2D's per-frame constants, float32, in the contract's order
    // (/root/reference/gsplat_plugin/shaders/GSplatShaderCoreLib.h:44-53)
    volatile float p00 = M4h(cam->proj, 0, 0), p11 = M4h(cam->proj, 1, 1);
    volatile float aspect = p00 / p11;
    volatile float tanFovX = 1.0f / p00;
    volatile float p11a = p11 * aspect;
    volatile float tanFovY = 1.0f / p11a;
    f->limx = 1.3f * tanFovX;
    f->limy = 1.3f * tanFovY;
    f->sigma_vo2 = (float)(sigma_max_sq(cam->view) * sigma_max_sq(cam->object) * 1.0001);
    f->W = (float)cam->width;
    f->H = (float)cam->height;
    volatile float wp = f->W * p00;
    f->focal = wp * 0.5f;
    f->width = cam->width;
    f->height = cam->height;
    f->sh_order = (c->has_sh && cam->sh_order > 0) ? (cam->sh_order > 3 ? 3 : cam->sh_order) : 0;
    f->tiles_x = (cam->width + GSR_TILE - 1) / GSR_TILE;
    f->tiles_y = (cam->height + GSR_TILE - 1) / GSR_TILE;
    f->shard_index = c->shard_index;
    f->shard_count = c->shard_count;
    f->shard_rpb = (c->shard_layout == 1 && c->shard_count > 1) ? (f->tiles_y + c->shard_count - 1) / c->shard_count : 0;
    if (f->shard_rpb > 0) {
        const int lo = c->shard_index * f->shard_rpb, hi = std::min(lo + f->shard_rpb, f->tiles_y);
        f->local_tiles_y = hi > lo ? hi - lo : 0;
    } else {
        f->local_tiles_y = (f->tiles_y > c->shard_index) ? (f->tiles_y - c->shard_index + c->shard_count - 1) / c->shard_count : 0;
    }
    // super-tile edge: smallest power of two that leaves <= 256 super-tiles (measured best at 1080p: S=8);
    // an explicit request is a lower bound -- the counting sort (k_binning.h) keeps one LDS bin per super-tile
    int shift = 0;
    if (c->opt_super > 0)
        while ((1 << shift) < c->opt_super) ++shift;
    while ((((f->tiles_x - 1) >> shift) + 1) * (((f->tiles_y - 1) >> shift) + 1) > 256) ++shift;
    // more than 256 tiles a side (> 4096 pixels): rects are packed in pairs of tiles (gsr_device.h), more than 512 (> 8192 pixels) in
    // blocks of four; a super-tile is then at least that wide (512 tiles a side cannot give <= 256 super-tiles otherwise), and its
    // edge in rect units stays <= 16: the list entries' column and row bits
    static_assert(GSR_MAX_DIM / GSR_TILE_PX == GSR_MAX_TILES_SIDE && GSR_MAX_TILES_SIDE <= 256 * 4, "tile coordinates are packed in 8 bits of rect units");
    f->rect_shift = (f->tiles_x > 512 || f->tiles_y > 512) ? 2 : ((f->tiles_x > 256 || f->tiles_y > 256) ? 1 : 0);
    if (shift < f->rect_shift) shift = f->rect_shift;
    f->super_shift = shift;
    f->flags = c->opt_flags;
    // sort-key range of this frame: distance^2 from cam_pos to the cloud's bounding box, as float bits
    f->key_min = 0u;
    f->key_max = 0xffffffffu;
    if (c->bbox_ok && !(c->opt_flags & GSR_FLAG_FULL_KEYS)) {
        double dmin2 = 0.0, dmax2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double p = cam->cam_pos[k];
            const double near_ = p < c->bb_lo[k] ? c->bb_lo[k] - p : (p > c->bb_hi[k] ? p - c->bb_hi[k] : 0.0);
            const double far_ = std::max(std::fabs(p - c->bb_lo[k]), std::fabs(p - c->bb_hi[k]));
            dmin2 += near_ * near_;
            dmax2 += far_ * far_;
        }
        const float lo = (float)(dmin2 * (1.0 - 1e-5)), hi = (float)(dmax2 * (1.0 + 1e-5));
        if (std::isfinite(lo) && std::isfinite(hi) && lo >= 0.0f && hi >= lo) {
            uint32_t blo, bhi;
            std::memcpy(&blo, &lo, 4);
            std::memcpy(&bhi, &hi, 4);
            f->key_min = blo > 2 ? blo - 2 : 0;
            f->key_max = bhi + 2;
        }
    }
    f->super = 1 << shift;
    f->stiles_x = ((f->tiles_x - 1) >> shift) + 1;
    f->stiles_y = ((f->tiles_y - 1) >> shift) + 1;
    int off = 0;
    for (int l = 0; l < GSR_PYR_LEVELS; ++l) {
        f->pyr_off[l] = off;
        off += gsr_pyr_dim(f->tiles_x, l) * gsr_pyr_dim(f->tiles_y, l);
    }
    f->cull_dilate = c->cull_pol.dilate;
    f->phase = 0;
}

// blockIdx -> tile table for the blend kernel.  Workgroup b lands on XCD b % 8; XCD x is
// handed the super-tiles x, x+8, x+16, ... (round robin for load balance), each with all of
// its owned tiles, so the 64 tiles that walk one list and gather the same records share an L2.
static int build_tile_map(gsr_context* c, const GsrFrame& f)
{
    if (c->map_w == f.width && c->map_h == f.height && c->map_si == c->shard_index && c->map_sc == c->shard_count &&
        c->map_rpb == f.shard_rpb && c->map_shift == f.super_shift && c->tile_map)
        return GSR_OK;
    int rc = sync_all(c);   // a frame in flight may still be reading the old table
    if (rc) return rc;
    const int n_super = f.stiles_x * f.stiles_y;
    std::vector<std::vector<int32_t>> per_xcd(8);
    for (int st = 0; st < n_super; ++st) {
        std::vector<int32_t>& v = per_xcd[st & 7];
        const int sx = st % f.stiles_x, sy = st / f.stiles_x;
        for (int gty = sy << f.super_shift; gty < ((sy + 1) << f.super_shift) && gty < f.tiles_y; ++gty) {
            const GsrShard sh{f.shard_index, f.shard_count, f.shard_rpb};
            if (!gsr_shard_owns(sh, gty)) continue;
            const int lty = f.shard_rpb > 0 ? gty - f.shard_index * f.shard_rpb : gty / c->shard_count;
            for (int tx = sx << f.super_shift; tx < ((sx + 1) << f.super_shift) && tx < f.tiles_x; ++tx)
                v.push_back(lty * f.tiles_x + tx);
        }
    }
    size_t chunk = 0;
    for (auto& v : per_xcd) chunk = std::max(chunk, v.size());
    std::vector<int32_t> map(chunk * 8 + 8, -1);
    for (int x = 0; x < 8; ++x)
        for (size_t k = 0; k < per_xcd[x].size(); ++k) map[k * 8 + x] = per_xcd[x][k];
    if (map.size() > c->map_cap) {
        dev_free(c->tile_map);
        c->map_cap = 0;
        rc = dev_alloc(&c->tile_map, map.size());
        if (rc) return rc;
        c->map_cap = map.size();
    }
    HIP_TRY(hipMemcpy(c->tile_map, map.data(), map.size() * 4, hipMemcpyHostToDevice));
    c->map_grid = (int)(chunk * 8);
    c->map_w = f.width; c->map_h = f.height; c->map_si = c->shard_index; c->map_sc = c->shard_count; c->map_rpb = f.shard_rpb;
    c->map_shift = f.super_shift;
    return GSR_OK;
}

static void harvest_slot(gsr_context* c, FrameSlot& sl)
{
    if (!sl.ev_pending) return;
    sl.ev_pending = false;
    hipEvent_t* e = sl.ev;
    if (hipEventSynchronize(e[6]) != hipSuccess) return;
    float ms[6] = {0, 0, 0, 0, 0, 0};
    if (!sl.ev_all) {   // only the blend kernel was bracketed
        if (hipEventElapsedTime(&ms[5], e[5], e[6]) != hipSuccess) ms[5] = 0.0f;
        c->st.ms_blend = ms[5];
        c->st.blend_ms_total += ms[5];
        c->st.blend_launches += 1;
        return;
    }
    for (int k = 0; k < 6; ++k)
        if (hipEventElapsedTime(&ms[k], e[k], e[k + 1]) != hipSuccess) ms[k] = 0.0f;
    c->st.ms_preprocess = ms[0];
    c->st.ms_depth_sort = ms[1];
    c->st.ms_emit = ms[2];
    c->st.ms_tile_sort = ms[3] + ms[4];
    c->st.ms_blend = ms[5];
    float tot = 0.0f;
    (void)hipEventElapsedTime(&tot, e[0], e[6]);
    c->st.ms_total = tot;
    c->st.blend_ms_total += ms[5];
    c->st.blend_launches += 1;
    c->st.frame_ms_total += tot;
    c->st.stage_ms_total[0] += ms[0];
    c->st.stage_ms_total[1] += ms[1];
    c->st.stage_ms_total[2] += ms[2];
    c->st.stage_ms_total[3] += ms[3] + ms[4];
    c->st.stage_ms_total[4] += ms[5];
    c->st.stage_frames += 1;
}

// ---------------------------------------------------------------------------
// A frame is QUEUED by frame_begin (everything, including a speculative back end) and COMPLETED by
// frame_finish (the one host-side wait of a frame: its pair count).  gsr_render = begin + finish; the
// multi-GPU driver begins on every GPU before it finishes on any, so the waits overlap; with
// GSR_OPT_DEFERRED_CHECK a device-target frame returns right after begin and its count is looked at later.
static inline int mark(FrameSlot& sl, int k)
{
    const FrameJob& j = sl.job;
    if (j.timing && (j.timing_all || k >= 5)) HIP_TRY(hipEventRecord(sl.ev[k], sl.stream));
    return GSR_OK;
}

// back end = placement + compositing.  The host needs the pair count D only to make sure the list buffer is
// large enough, so when a buffer exists the back end is queued SPECULATIVELY right behind the count (both
// kernels clamp to the buffer's capacity) and the host reads D while the GPU is already placing: the stream
// never drains mid-frame.  Only if D turns out to exceed the capacity (first frame, or the pair count grew by
// more than the 25 % headroom) is the buffer regrown and the back end run again.
static GsrRangeArgs range_args(gsr_context* c, FrameSlot& sl)
{
    const FrameJob& j = sl.job;
    GsrRangeArgs a;
    a.totals = sl.totals; a.n_super = j.n_super; a.sstart = sl.sstart; a.send = sl.send; a.host_total = sl.h_total_dev;
    a.ticket = j.ticket; a.max_pairs = (unsigned long long)GSR_MAX_PAIRS; a.redo_count = reinterpret_cast<uint32_t*>(sl.lazy_ctr);
    a.lazy_hint = c->lazy_hint; a.n_sorted = sl.d_n; a.k1_counts = sl.d_counts; a.sorted_keys = sl.keyA;
    a.depth_active = j.dcull ? sl.dactive + j.dpar : (const uint32_t*)nullptr;
    return a;
}

// the compositing launch (+ the lazy-colour fallback behind it)
static int queue_blend(gsr_context* c, FrameSlot& sl, bool with_depth, bool guarded)
{
    const FrameJob& j = sl.job;
    const GsrFrame& f = j.f;
    hipStream_t s = sl.stream;
    if (j.local_tiles > 0) {
        if (!j.direct) HIP_TRY(hipStreamWaitEvent(s, sl.ev_user, 0));
        GsrBlendArgs a;
        a.guard = guarded ? sl.dactive + j.dpar : (const uint32_t*)nullptr; a.guard_want = 0u;
        {   // (a perspective projection -- clip w = -view z, clip z = a z + b: window depth = (1 - a) / 2 - beta / view depth)
            const bool persp = f.pr[8] == 0.0f && f.pr[9] == 0.0f && f.pr[12] == 0.0f && f.pr[13] == 0.0f && f.pr[14] == -1.0f && f.pr[15] == 0.0f;
            a.near_alpha = persp ? 0.5f * (1.0f - f.pr[10]) : 0.0f;
            a.near_scale = persp ? 1.12f : 0.0f;
        }
        a.tile_cov = (with_depth && j.dcull && !j.dblind) ? sl.dpyr + 4 * sl.dpyr_cap : (const float*)nullptr;
        a.idx_mask = f.idx_mask; a.zq0 = f.zq0; a.zqs = f.zqs;
        a.tile_dmax = (with_depth && j.dcull && !j.dblind && f.idx_mask != 0xffffffffu) ? sl.dpyr + (size_t)(2 * j.dpar) * sl.dpyr_cap + f.pyr_off[0] : (const float*)nullptr;
        a.width = f.width; a.height = f.height; a.tiles_x = f.tiles_x; a.local_tiles = j.local_tiles;
        a.shard = GsrShard{f.shard_index, f.shard_count, f.shard_rpb}; a.band_rows = j.band_rows;
        a.super_shift = f.super_shift; a.rect_shift = f.rect_shift; a.stiles_x = f.stiles_x; a.use_map = j.use_map ? 1 : 0; a.flags = f.flags;
        a.list_cap = (int32_t)std::min<size_t>(sl.pair_cap, (size_t)0x7fffffff);
        a.sup_work = (a.use_map && c->opt_swizzle >= 2 && sl.sup_work) ? sl.sup_work + 256 * sl.sup_par : nullptr;
        a.slab = j.phase; a.tbuf = sl.tbuf; a.tile_work_a = sl.tile_work_a;
        uint4* const tw = j.phase == 1 ? sl.tile_work_a : sl.tile_work;   // (phase 1's bookkeeping is kept for phase 2 and the frame end)
        // heaviest-first table of this slot's previous frame, if that frame had the same tiles
        const int sig[6] = {f.width, f.height, f.shard_index, f.shard_count, f.shard_rpb, f.super_shift};
        const bool ordered = a.use_map && c->opt_swizzle >= 2 && sl.order_valid && std::memcmp(sig, sl.order_sig, sizeof sig) == 0;
        const int32_t* tmap = ordered ? sl.order : c->tile_map;
        const unsigned grid = ordered ? (unsigned)(8 * sl.order_per_xcd) : a.use_map ? (unsigned)c->map_grid : (unsigned)j.local_tiles;
        GsrLazyArgs lz;
        lz.f = f; lz.colrow = c->colrow;
        lz.redo = j.lazy ? sl.redo : nullptr;
        lz.redo_count = reinterpret_cast<uint32_t*>(sl.lazy_ctr);
        float4* tgt = reinterpret_cast<float4*>(j.target);
        // Host-target frames (a caller without GL interop): the frame's LAST blend launch is issued band by band of tile rows, an event
        // behind each, and queue_frame_end copies a band's rows back on a second stream as soon as its event fires: all but the first
        // band's compositing hides behind the link, which is what bounds such a frame (33 MB at ~55 GB/s = 0.6 ms per 1080p frame).
        // The launches walk the same tile table; a workgroup of another band's tile leaves at once (~3 us per extra launch).
        int nb = 1;
        if (!j.out_is_device && j.phase != 1 && c->opt_host_bands > 1 && c->shard_count == 1 && f.local_tiles_y >= 4 * c->opt_host_bands) nb = c->opt_host_bands;
        if (nb > 1 && !c->copy_stream && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { c->copy_stream = nullptr; nb = 1; (void)hipGetLastError(); }
        for (int b = 0; b < nb && nb > 1; ++b)
            if (!sl.ev_band[b] && hipEventCreateWithFlags(&sl.ev_band[b], hipEventDisableTiming) != hipSuccess) { sl.ev_band[b] = nullptr; nb = 1; (void)hipGetLastError(); }
        sl.bands = nb;
        for (int b = 0; b < nb; ++b) {
            a.row_lo = nb > 1 ? (int)((int64_t)f.local_tiles_y * b / nb) : 0;
            a.row_hi = nb > 1 ? (int)((int64_t)f.local_tiles_y * (b + 1) / nb) : 0x7fffffff;
            sl.band_row[b] = a.row_lo; sl.band_row[b + 1] = nb > 1 ? a.row_hi : f.local_tiles_y;
            if (with_depth)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_blend<true>), dim3(grid), dim3(256), 0, s, a, tmap, sl.pvA, sl.sstart,
                                   sl.send, sl.rec, tgt, tw, sl.zwin, j.d_depth, lz);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_blend<false>), dim3(grid), dim3(256), 0, s, a, tmap, sl.pvA, sl.sstart,
                                   sl.send, sl.rec, tgt, tw, sl.zwin, j.d_depth, lz);
            if (j.lazy) {   // the tiles that met a pending colour, with on-demand evaluation (normally none: the blocks exit at once)
                if (with_depth)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_blend_lazy<true>), dim3((unsigned)j.local_tiles), dim3(256), 0, s, a, c->tile_map,
                                       sl.pvA, sl.sstart, sl.send, sl.rec, tgt, sl.tile_work, sl.zwin, j.d_depth, lz);
                else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_blend_lazy<false>), dim3((unsigned)j.local_tiles), dim3(256), 0, s, a, c->tile_map,
                                       sl.pvA, sl.sstart, sl.send, sl.rec, tgt, sl.tile_work, sl.zwin, j.d_depth, lz);
            }
            if (nb > 1) HIP_TRY(hipEventRecord(sl.ev_band[b], s));
        }
        HIP_TRY(hipGetLastError());
    }
    return GSR_OK;
}

static int queue_back_end(gsr_context* c, FrameSlot& sl)
{
    const FrameJob& j = sl.job;
    const GsrFrame& f = j.f;
    hipStream_t s = sl.stream;
    int rc;
    if ((rc = mark(sl, 3))) return rc;
    if (j.n > 0) {
        const uint32_t nblk = div_up(j.n, (uint32_t)BN_THREADS * (uint32_t)j.bn_items);
        const size_t lds = (size_t)4 * j.bn_items * j.n_super * (8 + 4);   // lane masks (u64) + first list position (u32) per (group of 64 splats, super-tile)
        const GsrShard shd{f.shard_index, f.shard_count, f.shard_rpb, f.rect_shift};
        const GsrRangeArgs ra = j.ranges_folded ? range_args(c, sl) : GsrRangeArgs{};
        // (+ the extra work items of the blocks that are split by rows of super-tiles, k_binning.h; +1: the publishing workgroup)
        const uint32_t bn_extra = (uint32_t)BN_SPLIT_TILES * (uint32_t)std::max(f.stiles_y - 1, 0);
        const uint32_t grid = std::min(nblk, j.bn_grid) + bn_extra + (j.ranges_folded ? 1u : 0u);
#define GSR_PLACE(I) hipLaunchKernelGGL((f.idx_mask != 0xffffffffu ? k_bin_place<I, true> : k_bin_place<I, false>), dim3(grid), dim3(BN_THREADS), lds, s, sl.valA, sl.d_n,          \
                                        f.super_shift - f.rect_shift, shd, f.stiles_x, j.n_super, sl.hist, sl.sstart, nblk, (uint32_t)sl.pair_cap, sl.pvA, ra,   \
                                        f.idx_mask != 0xffffffffu ? sl.zwin : (const float*)nullptr, f.zq0, f.zqs)
        if (j.bn_items == 1) GSR_PLACE(1); else if (j.bn_items == 2) GSR_PLACE(2); else GSR_PLACE(4);
#undef GSR_PLACE
        HIP_TRY(hipGetLastError());
    }
    if ((rc = mark(sl, 4))) return rc;
    if (j.lazy && j.n > 0) {
        // colours for the front of every super-tile list: as deep as the previous frame's tiles scanned (+ headroom)
        const bool predict = c->prefix_valid && !(f.flags & GSR_FLAG_LAZY_NO_PREFIX);
        hipLaunchKernelGGL(k_colour_prefix, dim3((unsigned)(j.n_super * CL_BLOCKS_PER_LIST)), dim3(CL_THREADS), 0, s, f, sl.pvA,
                           sl.sstart, sl.send, (f.flags & GSR_FLAG_LAZY_NO_PREFIX) ? c->prefix_none : (predict ? c->prefix : c->prefix_all),
                           (int)std::min<size_t>(sl.pair_cap, (size_t)0x7fffffff), c->colrow, sl.rec, sl.colour_evals);
        HIP_TRY(hipGetLastError());
    }
    if ((rc = mark(sl, 5))) return rc;
    // Depth-tested frames: the plain kernel, guarded, while the slot's depth buffers have been clear (k_blend.h: GsrBlendArgs.guard);
    // not for a frame that is handed over before its mailbox is read (nobody could queue the other kernel in time)
    // (... nor over lists whose index words carry depth codes: only the depth-tested kernel masks them off)
    sl.job.blend_guess_plain = j.d_depth != nullptr && j.dcull && !c->depth_active && !j.deferred && j.f.idx_mask == 0xffffffffu;
    if ((rc = queue_blend(c, sl, j.d_depth != nullptr && !sl.job.blend_guess_plain, sl.job.blend_guess_plain))) return rc;
    return mark(sl, 6);
}

// bookkeeping + hand-over: the frame's counters to the host mirror, host copy if asked, result ordered on the public stream
static int queue_frame_end(gsr_context* c, FrameSlot& sl)
{
    const FrameJob& j = sl.job;
    hipStream_t s = sl.stream;
    GsrSumArgs g;
    g.n_tiles = j.local_tiles; g.tiles_x = j.f.tiles_x; g.tiles_y = j.f.tiles_y; g.shard = GsrShard{j.f.shard_index, j.f.shard_count, j.f.shard_rpb};
    g.super_shift = j.f.super_shift; g.stiles_x = j.f.stiles_x; g.n_super = j.n_super;
    GsrHorizonArgs hz{};
    hz.list_cap = (int32_t)std::min<size_t>(sl.pair_cap, (size_t)0x7fffffff);
    // (below a few hundred thousand splats in the sort the frame is bound by launch floors: nothing for culling to win)
    // ... and while culling is held off nobody needs horizons: they are prepared again two frames before it may resume.
    // A deferred frame (handed over before its pair count was known) never culls and may have clamped lists: no horizons from it.
    if (c->opt_cull && c->opt_cull != 3 && j.n > 0 && !j.deferred && (c->opt_cull >= 2 || j.cull || j.phase == 2 || (c->cull_pol.vis_unculled >= 300000u && c->cull_pol.holdoff <= 2))) {
        hz.raw = sl.hraw; hz.pyr_in = sl.hpyr; hz.pyr_out = sl.hpyr_next; hz.dilate = j.f.cull_dilate; hz.culled = j.cull ? 1 : 0; hz.lists = sl.pvA; hz.geoA = c->geoA;
        for (int l = 0; l < GSR_PYR_LEVELS; ++l) hz.pyr_off[l] = j.f.pyr_off[l];
        hz.cam[0] = j.f.cam[0]; hz.cam[1] = j.f.cam[1]; hz.cam[2] = j.f.cam[2];
        hz.host_end = j.cull ? sl.h_end_dev : nullptr; hz.ticket = j.ticket;
        // (depth-tested frames leave, and culled ones check, the tiles' status: k_blend.h.  A frame without a depth buffer leaves every
        //  tile classic: the sign bits of its raw horizons are clear)
        hz.stat = sl.hstat;
        hz.stat_in_use = (j.cull && j.dcull && j.dstat) ? 1 : 0;
        hz.depth_culled = (j.dcull && !j.dblind) ? 1 : 0;
        hz.idx_mask = j.f.idx_mask;
        static const bool dbgv = std::getenv("GSR_DEBUG_VIOL") != nullptr;
        if (dbgv) {
            if (!sl.dbg_viol && hipMalloc(reinterpret_cast<void**>(&sl.dbg_viol), 80 * 4) != hipSuccess) sl.dbg_viol = nullptr;
            if (sl.dbg_viol) (void)hipMemsetAsync(sl.dbg_viol, 0, 80 * 4, s);
            hz.dbg = sl.dbg_viol;
        }
    }
    if (j.phase == 2) { hz.slab = 2; hz.tile_work_a = sl.tile_work_a; }
    const int nblocks8 = ((j.f.tiles_x + 7) >> 3) * ((j.f.tiles_y + 7) >> 3);
    hipLaunchKernelGGL(k_tile_pass, dim3((unsigned)nblocks8), dim3(64), 0, s, sl.tile_work, g, sl.sstart, sl.send, hz, sl.partial, sl.st_scan);
    uint32_t* const prefix_arg = (j.f.sh_order > 0 && c->opt_lazy) ? c->prefix : (uint32_t*)nullptr;
    const uint32_t* const redo_arg = j.lazy ? reinterpret_cast<const uint32_t*>(sl.lazy_ctr) : (const uint32_t*)nullptr;
    uint32_t* const work_next = sl.sup_work ? sl.sup_work + 256 * (sl.sup_par ^ 1) : (uint32_t*)nullptr;
    sl.horizon_valid = hz.raw != nullptr;
    sl.hstat_valid = hz.raw != nullptr;       // (the dilation writes it beside the pyramid)
    // The heaviest-first table is rebuilt every frame where the tiles differ a lot (k_sum_work's verdict); where they do not it is
    // still worth a few microseconds of k_blend's tail (C4: 0.142 -> 0.139 ms), but not the 10 us of k_tile_order every frame:
    // then a table stands for GSR_ORDER_KEEP frames of the same shape.
    const int osig[6] = {j.f.width, j.f.height, j.f.shard_index, j.f.shard_count, j.f.shard_rpb, j.f.super_shift};
    const bool order_due = c->opt_swizzle >= 3 || (c->opt_swizzle == 2 && c->order_pays);
    const bool order_kept = !order_due && c->opt_swizzle == 2 && c->opt_order_keep > 0 && sl.order_valid && sl.order_age < c->opt_order_keep &&
                            std::memcmp(osig, sl.order_sig, sizeof osig) == 0;
    const bool order_refresh = !order_due && !order_kept && c->opt_swizzle == 2 && c->opt_order_keep > 0 && c->n >= 1000000u;
    if (order_kept) sl.order_age += 1; else sl.order_valid = false;
    const bool order_now = j.use_map && (order_due || order_refresh) && j.local_tiles > 0 && j.n_super <= 256 && sl.sup_work;
    const int per_xcd = 2 * ((j.n_super + 15) / 16) << (2 * j.f.super_shift);
    if (order_now) {
        const size_t want = (size_t)8 * per_xcd;
        if (want > sl.order_cap) {
            HIP_TRY(hipStreamSynchronize(s));
            dev_free(sl.order);
            sl.order_cap = 0;
            int rc = dev_alloc(&sl.order, want);
            if (rc) return rc;
            sl.order_cap = want;
        }
    }
    // the next frame's tile order is built BESIDE the frame's sums (and the horizon dilation), in the same launch (k_blend.h: k_frame_end_order)
    const bool fused_order = order_now && c->opt_fuse_order != 0;
    if (sl.horizon_valid) sl.hpyr_re = std::min(c->cull_pol.dilate, GSR_DILATE_EXACT_MAX);
    if (fused_order) {
        const GsrOrderArgs oa{sl.tile_work, j.f.tiles_y, sl.sup_work + 256 * sl.sup_par, per_xcd, sl.order};
        const int ndil = sl.horizon_valid ? nblocks8 : 0;
        hipLaunchKernelGGL(k_frame_end_order, dim3(9u + (unsigned)((ndil + TO_THREADS / 64 - 1) / (TO_THREADS / 64))), dim3(TO_THREADS), 0, s, sl.partial, nblocks8, ndil, g, sl.counters, sl.d_n, sl.d_frame,
                           prefix_arg, redo_arg, sl.colour_evals, sl.lazy_ctr + 1, sl.sstart, sl.send, c->lazy_hint, work_next, hz, sl.st_scan, sl.hpyr_re, oa);
        HIP_TRY(hipGetLastError());
    } else if (sl.horizon_valid) {
        // the frame's sums and verdict, and BESIDE them (one launch) the next frame's pyramid: every tile's horizon widened to its
        // neighbourhood, into the slot's other pyramid buffer
        hipLaunchKernelGGL(k_frame_end, dim3((unsigned)nblocks8 + 1u), dim3(SW_THREADS), 0, s, sl.partial, nblocks8, g, sl.counters, sl.d_n, sl.d_frame,
                           prefix_arg, redo_arg, sl.colour_evals, sl.lazy_ctr + 1, sl.sstart, sl.send, c->lazy_hint, work_next, hz, sl.st_scan, sl.hpyr_re);
        HIP_TRY(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_sum_work, dim3(1), dim3(SW_THREADS), 0, s, sl.partial, nblocks8, g, sl.counters, sl.d_n, sl.d_frame,
                           prefix_arg, redo_arg, sl.colour_evals, sl.lazy_ctr + 1, sl.sstart, sl.send, c->lazy_hint, work_next, hz, sl.st_scan);
        HIP_TRY(hipGetLastError());
    }
    if (sl.horizon_valid) {
        std::swap(sl.hpyr, sl.hpyr_next);      // (what the slot's next frame culls against)
        const int sig[7] = {j.f.width, j.f.height, j.f.shard_index, j.f.shard_count, j.f.shard_rpb, j.f.super_shift, (int)c->geo_gen};
        std::memcpy(sl.horizon_sig, sig, sizeof sig);
        sl.horizon_cam = j.cam_arg;
    }
    if (order_now) {
        if (!fused_order) {
            hipLaunchKernelGGL(k_tile_order, dim3(8), dim3(TO_THREADS), 0, s, sl.tile_work, g, j.f.tiles_y, sl.sup_work + 256 * sl.sup_par,
                               per_xcd, sl.order);
            HIP_TRY(hipGetLastError());
        }
        std::memcpy(sl.order_sig, osig, sizeof osig);
        sl.order_per_xcd = per_xcd;
        sl.order_valid = true;
        sl.order_age = 0;
    }
    sl.sup_par ^= 1;
    if (j.f.sh_order > 0 && c->opt_lazy) c->prefix_valid = j.phase == 0;   // (eager frames keep the scan depths too: the switch to lazy starts predicted;
                                                                           //  a front-slab frame's scan depths belong to two different lists)
    sl.last_lazy = j.lazy;
    if (j.timing) { sl.ev_pending = true; sl.ev_all = j.timing_all; }
    if (!j.out_is_device) {
        if (sl.bands > 1 && c->copy_stream) {
            // (band by band behind the blend launches' events, on a stream of their own: k_tile_pass and the frame end run beside the copies)
            for (int b = 0; b < sl.bands; ++b) {
                const size_t r0 = std::min<size_t>((size_t)sl.band_row[b] * GSR_TILE, (size_t)j.band_rows), r1 = std::min<size_t>((size_t)sl.band_row[b + 1] * GSR_TILE, (size_t)j.band_rows);
                if (r1 <= r0) continue;
                const size_t off = r0 * (size_t)j.f.width * 4;
                HIP_TRY(hipStreamWaitEvent(c->copy_stream, sl.ev_band[b], 0));
                HIP_TRY(hipMemcpyAsync(j.user_out + off, sl.fb + off, (r1 - r0) * (size_t)j.f.width * 16, hipMemcpyDeviceToHost, c->copy_stream));
            }
            HIP_TRY(hipStreamSynchronize(c->copy_stream));
            HIP_TRY(hipStreamSynchronize(s));
        } else {
            HIP_TRY(hipMemcpyAsync(j.user_out, sl.fb, j.out_px * 16, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
    }
    // results are ordered on the public stream: anything the caller queues there next sees this frame
    if (!j.direct) {
        HIP_TRY(hipEventRecord(sl.ev_done, s));
        HIP_TRY(hipStreamWaitEvent(c->stream, sl.ev_done, 0));
    }
    return GSR_OK;
}

// The middle of a front-slab frame: phase 1 is composited; which tiles are finished?  (k_blend.h: k_slab_mid, then the pyramid
// phase 2 culls against -- no dilation: same frame, same camera)
static int queue_slab_mid(gsr_context* c, FrameSlot& sl)
{
    const FrameJob& j = sl.job;
    hipStream_t s = sl.stream;
    GsrSumArgs g;
    g.n_tiles = j.local_tiles; g.tiles_x = j.f.tiles_x; g.tiles_y = j.f.tiles_y; g.shard = GsrShard{j.f.shard_index, j.f.shard_count, j.f.shard_rpb};
    g.super_shift = j.f.super_shift; g.stiles_x = j.f.stiles_x; g.n_super = j.n_super;
    GsrHorizonArgs hz{};
    hz.list_cap = (int32_t)std::min<size_t>(sl.pair_cap, (size_t)0x7fffffff);
    hz.raw = (c->opt_cull && c->opt_cull != 3) ? sl.hraw : nullptr;          // the finished tiles' horizons for the slot's NEXT frame
    hz.lists = sl.pvA; hz.geoA = c->geoA; hz.idx_mask = j.f.idx_mask;
    for (int l = 0; l < GSR_PYR_LEVELS; ++l) hz.pyr_off[l] = j.f.pyr_off[l];
    hz.cam[0] = j.f.cam[0]; hz.cam[1] = j.f.cam[1]; hz.cam[2] = j.f.cam[2];
    const int nblocks8 = ((j.f.tiles_x + 7) >> 3) * ((j.f.tiles_y + 7) >> 3);
    hipLaunchKernelGGL(k_slab_mid, dim3((unsigned)nblocks8), dim3(64), 0, s, sl.tile_work_a, g, sl.sstart, sl.send, hz, sl.hpyr2,
                       sl.slab + GSR_SLAB_BINS, j.f.key_min);
    HIP_TRY(hipGetLastError());
    return GSR_OK;
}

// a frame that failed half-way may have kernels queued that still write the caller's buffer: drain them first
static int frame_abort(FrameSlot& sl, int rc)
{
    sl.job.open = false;
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    return rc;
}

// A mailbox word in mapped host memory carries (frame ticket << 32 | value); the GPU writes it mid-stream while it carries on
// -- no event in the stream (an event costs the GPU ~6 us of idle queue), no API call.  The host watches the word: a bounded
// spin (the word normally arrives within tens of microseconds), then it stops burning the caller's core and blocks in
// hipStreamSynchronize -- by then the frame is a long one and the extra latency no longer matters.
static int wait_mailbox(FrameSlot& sl, volatile unsigned long long* box, uint32_t ticket, const char* what, unsigned long long* out)
{
    unsigned long long v = *box;
    if ((uint32_t)(v >> 32) != ticket) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned long spins = 1; (uint32_t)(v >> 32) != ticket; ++spins) {
            if ((spins & 0xffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(GSR_SPIN_US)) {
                const hipError_t q = hipStreamSynchronize(sl.stream);   // everything queued has run: the word must be there now
                if (q != hipSuccess) return set_err(GSR_E_HIP, "gsr_render: waiting for the %s: %s", what, hipGetErrorString(q));
                v = *box;
                if ((uint32_t)(v >> 32) != ticket) return set_err(GSR_E_HIP, "gsr_render: the frame finished without delivering its %s", what);
                break;
            }
            __builtin_ia32_pause();
            v = *box;
        }
    }
    *out = v;
    return GSR_OK;
}

// Did the camera JUMP since the frame that left the slot's depth horizons (a cut, a viewport switch, "home")?  Horizons tolerate a
// view that moves by a few tiles per frame (they are widened by the dilation radius, 16 tiles at most) and distances that change by
// a few per cent (a horizon sits a quarter of the tile's depth range behind what the tile needed).  Beyond that a culled attempt is
// wasted -- its check fails and the frame is rendered again -- so such a frame is rendered as if the slot had no horizons (a
// front-slab frame where that pays).  A heuristic on nine reference points (the corners and the centre of the cloud's bounding
// box: their direction as seen from the camera, and their distance); whatever it answers, the pixels are the same.
static bool camera_jumped(const gsr_context* c, const gsr_camera& was, const gsr_camera& now)
{
    if (!c->bbox_ok) return false;
    double diag2 = 0.0;
    for (int k = 0; k < 3; ++k) diag2 += ((double)c->bb_hi[k] - c->bb_lo[k]) * ((double)c->bb_hi[k] - c->bb_lo[k]);
    if (!(diag2 > 0.0) || !std::isfinite(diag2)) return false;
    const double near2 = diag2 / 16.0;                              // reference points closer than a quarter of the diagonal say nothing
    const double focal_px = 0.5 * std::fabs((double)now.proj[0]) * now.width;
    const double max_px = 24.0 * GSR_TILE_PX;
    int used = 0;
    for (int k = 0; k < 9; ++k) {
        double p[3], va[3], vb[3], da2 = 0.0, db2 = 0.0;
        for (int a = 0; a < 3; ++a) p[a] = k < 8 ? (((k >> a) & 1) ? c->bb_hi[a] : c->bb_lo[a]) : 0.5 * ((double)c->bb_lo[a] + c->bb_hi[a]);
        for (int r = 0; r < 3; ++r) {                               // view-space positions (GL column-major: m[c * 4 + r])
            va[r] = was.obj_view[r] * p[0] + was.obj_view[4 + r] * p[1] + was.obj_view[8 + r] * p[2] + was.obj_view[12 + r];
            vb[r] = now.obj_view[r] * p[0] + now.obj_view[4 + r] * p[1] + now.obj_view[8 + r] * p[2] + now.obj_view[12 + r];
            da2 += va[r] * va[r]; db2 += vb[r] * vb[r];
        }
        if (!(da2 > near2) || !(db2 > near2) || !std::isfinite(da2) || !std::isfinite(db2)) continue;
        ++used;
        const double ratio2 = db2 / da2;
        if (ratio2 > 1.25 * 1.25 || ratio2 < 0.8 * 0.8) return true;
        const double cosang = (va[0] * vb[0] + va[1] * vb[1] + va[2] * vb[2]) / std::sqrt(da2 * db2);
        const double ang = std::acos(std::min(1.0, std::max(-1.0, cosang)));
        if (ang * focal_px > max_px) return true;
    }
    (void)used;
    return false;
}

static int frame_begin(gsr_context* c, const gsr_camera* cam, const float* depth, int depth_is_device,
                       float* rgba_out, int out_is_device, FrameSlot** used, bool allow_cull, int phase_in = 0);

static int frame_finish(gsr_context* c, FrameSlot& sl)
{
    if (!sl.job.open) return GSR_OK;
    FrameJob& j = sl.job;
    HIP_TRY(hipSetDevice(c->device));
    uint32_t D = 0;
    if (j.n > 0) {
        // The pair count arrives in mapped host memory, stamped with the frame's ticket, while the GPU carries on: the
        // host just watches the word -- no event in the stream (an event costs the GPU ~6 us of idle queue), no API call.
        volatile unsigned long long* box = sl.h_total;
        unsigned long long v = 0;
        const int wrc = wait_mailbox(sl, box, j.ticket, "pair count", &v);
        if (wrc) return frame_abort(sl, wrc);
        D = (uint32_t)v;
        j.sort_failed = j.local_sort && (box[1] & 32ull) != 0ull;
        if (j.sort_failed) {
            // the small-frame sort gave a bucket up: this frame's order, lists and pair count mean nothing.  frame_check renders it
            // again (three global passes); nothing of it is kept -- not the order (sort cache), not its bookkeeping (no frame end),
            // not its say in the policies below -- and the work sums its speculative back end added over garbage lists are cleared
            sl.sort_valid = false;
            sl.kept_hint = 0; sl.kept_lo = sl.kept_hi = 0;
            sl.last_pairs = 0;
            if (sl.sup_work) (void)hipMemsetAsync(sl.sup_work + 256 * sl.sup_par, 0, 256 * sizeof(uint32_t), sl.stream);
            sl.local_pol.on_sort_result(true);       // (back-off: GsrLocalSortPolicy::begin_frame)
            j.open = false;
            return GSR_OK;
        }
        if (j.local_sort) sl.local_pol.on_sort_result(false);
        if (j.dcull && j.dblind && j.cull && (box[1] & 64ull) != 0ull) {
            // The frame was culled against horizons, its depth buffer turned out to hold opaque geometry, and k_cluster_cull did not know
            // (the pyramids were built beside it: the previous depth buffer was clear).  Horizons speak for uncovered pixels only; the
            // covered ones are served by the depth clause, which k_cluster_cull could not apply: clusters they need may be gone.  The
            // frame again, with the pyramids in front (once, when opaque geometry first appears; frame_check).
            c->depth_active = true;
            c->st.frames_requeued += 1;
            if (sl.sup_work) (void)hipMemsetAsync(sl.sup_work + 256 * sl.sup_par, 0, 256 * sizeof(uint32_t), sl.stream);
            sl.sort_valid = false;
            j.redo = true;
            j.open = false;
            return GSR_OK;
        }
        if (j.dcull) {   // (the kernels' word on THIS frame's depth buffer: does it hold anything in front of the far plane?)
            c->depth_active = (box[1] & 64ull) != 0ull;
            if (j.sort_fresh) sl.sorted_dculled = c->depth_active;   // (conservatively "yes" until now)
        }
        if (j.phase != 2) {   // (the kernels' verdicts on the frame BEFORE: lazy colour / occlusion culling / list prefixes pay)
            c->lazy_pays = (box[1] & 1ull) != 0ull;
            c->cull_pol.on_kernel_verdict((box[1] & 4ull) != 0ull);
            c->prefix_cheaper = (box[1] & 8ull) != 0ull;
            c->order_pays = (box[1] & 2ull) != 0ull;
        }
        if (j.phase == 0) {
        {   // occlusion culling earns its keep only if it drops a good part of what an unculled frame keeps
            const uint32_t kept = (uint32_t)(box[1] >> 32);
            c->cull_pol.on_kept(j.cull, kept, c->opt_cull);
        }
        sl.surv_hint = (uint32_t)box[2];           // clusters that survived k_cluster_cull: sizes the next frame's K1 grid
        sl.kept_hint = (uint32_t)(box[1] >> 32);   // ... and how many splats reached the depth sort: picks the next frame's sort
        sl.kept_culled = j.cull;
        } else {
            // a front-slab frame: its two phases say nothing about what an ordinary frame keeps (no say in the policies); the next
            // frame -- usually one culled against this frame's horizons -- keeps about what both phases kept, from about as many clusters
            const uint32_t kept = (uint32_t)(box[1] >> 32);
            static const bool dbg = std::getenv("GSR_SLAB_DEBUG") != nullptr;
            if (dbg) fprintf(stderr, "[slab] phase %d: clusters through K1 %u, splats kept %u, pairs %u\n", j.phase, (uint32_t)box[2], kept, D);
            if (j.phase == 1) { sl.slab_kept = kept; sl.slab_kept1 = std::max(kept, 1u); }
            else {
                sl.slab_kept2 = std::max(kept, 1u);
                sl.kept_hint = sl.slab_kept + kept;
                // a slab behind which most of the frame still has to be drawn (a ball seen from afar: its near cap finishes few tiles;
                // oblique ground; a wall) costs more than it saves
                c->slab_pol.on_frame_done(sl.kept_hint, c->cull_pol.vis_unculled);
                sl.surv_hint = std::min<uint32_t>((uint32_t)box[2], std::max<uint32_t>(16384u, 2u * div_up(sl.kept_hint, GSR_CLUSTER)));
                sl.kept_culled = true;
            }
        }
        if (sl.kept_hint > 0 && j.phase == 0) {    // ... between which keys (stored relative to THIS frame's key_min)
            sl.kept_lo = (uint32_t)box[3] + j.f.key_min;
            sl.kept_hi = (uint32_t)(box[3] >> 32) + j.f.key_min;
        } else {
            sl.kept_lo = sl.kept_hi = 0;
        }
        if (D == 0xffffffffu || (unsigned long long)D > (unsigned long long)GSR_MAX_PAIRS)
            return frame_abort(sl, set_err(GSR_E_TOO_MANY_PAIRS, "gsr_render: the frame's super-tile pairs exceed the limit of %lld", GSR_MAX_PAIRS));
        // (a slot that has never met a pair has no list buffer at all: the frame-end kernels read entry 0 of it unconditionally --
        //  a rank whose rows see nothing, under forced culling or forced front-slab frames, faulted on the null pointer)
        const bool short_buffer = D > sl.pair_cap || !sl.pvA;
        if (short_buffer) {
            (void)hipStreamSynchronize(sl.stream);   // a speculative (clamped) back end may still be reading the old buffer
            const bool first_buffer = sl.pvA == nullptr;
            dev_free(sl.pvA);
            sl.pair_cap = 0;
            // (a slot's first buffer may be sized by a frame -- or a front slab -- that shows next to nothing: at least two entries per splat
            //  then, up to 32 MB, so that the frames behind it need not each regrow it)
            const size_t floor_ = !first_buffer ? 0 : std::min<size_t>((size_t)2 * j.n + 4096, (size_t)1 << 22);
            const size_t want = std::max<size_t>((size_t)D + D / 4 + 4096, std::max(floor_, c->pair_want));
            int rc = dev_alloc(&sl.pvA, want + 4);   // (+4: the blend kernel scans in 4-entry steps)
            if (rc) return frame_abort(sl, rc);
            sl.pair_cap = want;
            c->pair_want = std::max(c->pair_want, want);   // the other frame slot grows before its next frame
        }
        if (short_buffer && j.speculative && j.phase == 2) {
            // Phase 2 of a front-slab frame CONTINUES from what phase 1 left in the target: the clamped back end that has run has
            // composited on top of it, and a second run would composite the same splats again.  The whole frame again (frame_check),
            // with the buffer that now fits; nothing of this attempt is kept.
            c->st.frames_requeued += 1;
            if (sl.sup_work) (void)hipMemsetAsync(sl.sup_work + 256 * sl.sup_par, 0, 256 * sizeof(uint32_t), sl.stream);
            sl.sort_valid = false;
            j.redo = true;
            j.open = false;
            return GSR_OK;
        }
        if (j.deferred) {
            // the frame was handed over before its pair count was known; if the lists were clamped it misses their tails
            if (short_buffer) c->st.frames_truncated += 1;
        } else if (short_buffer || !j.speculative) {
            if (short_buffer && j.speculative) {
                c->st.frames_requeued += 1;
                // the first, clamped back end has already added its tiles' work to the sums k_tile_order reads
                if (sl.sup_work) (void)hipMemsetAsync(sl.sup_work + 256 * sl.sup_par, 0, 256 * sizeof(uint32_t), sl.stream);
            }
            int rc = queue_back_end(c, sl);
            if (rc) return frame_abort(sl, rc);
        }
    }
    sl.last_pairs = D;
    if (j.n > 0 && j.blend_guess_plain && (sl.h_total[1] & 64ull) != 0ull) {
        // the guarded plain kernel found covered pixels and did nothing: the depth-tested one draws the frame
        j.blend_guess_plain = false;
        int rc = queue_blend(c, sl, true, false);
        if (rc) return frame_abort(sl, rc);
    }
    if (j.phase == 1) {
        // front slab composited: find the finished tiles, then the rest of the frame (same slot, same call arguments)
        int rc = queue_slab_mid(c, sl);
        if (rc) return frame_abort(sl, rc);
        if (j.timing) { sl.ev_pending = true; sl.ev_all = false; }
        j.open = false;
        const gsr_camera cam = j.cam_arg;
        const float* depth = j.depth_arg;
        const int ddev = j.depth_is_device_arg, odev = j.out_is_device_arg;
        float* out = j.out_arg;
        c->frame_no -= 1;          // (the same frame, in the same slot)
        c->st.frames -= 1;
        FrameSlot* sl2 = nullptr;
        rc = frame_begin(c, &cam, depth, ddev, out, odev, &sl2, false, 2);
        if (rc) return rc;
        return frame_finish(c, *sl2);
    }
    if (!j.deferred) {
        int rc = queue_frame_end(c, sl);
        if (rc) return frame_abort(sl, rc);
    }
    j.open = false;
    return GSR_OK;
}

// Occlusion culling: wait for k_sum_work's word on the frame just queued (it is the first thing that kernel writes, so the
// host is back in time to queue the next frame behind it): did a tile run off a list cut at its horizon?
static int frame_verdict(gsr_context* c, FrameSlot& sl, bool* broke)
{
    (void)c;
    const FrameJob& j = sl.job;
    volatile unsigned long long* box = sl.h_end;
    unsigned long long v = 0;
    const int wrc = wait_mailbox(sl, box, j.ticket, "culling verdict", &v);
    if (wrc) return wrc;
    *broke = (v & 1ull) != 0ull;
    return GSR_OK;
}

static int finish_open_frames(gsr_context* c)
{
    int rc = GSR_OK;
    for (int k = 0; k < GSR_MAX_SLOTS; ++k) {
        const int r = frame_finish(c, c->slot[k]);
        if (r && !rc) rc = r;
    }
    return rc;
}

// the depth order of ALL splats for the camera position of frame f (k_sort.h): keys -> LSD passes -> c->pos_order
static int build_pos_order(gsr_context* c, FrameSlot& sl, const GsrFrame& f)
{
    const uint32_t n = c->n;
    int rc;
    uint32_t *kA = nullptr, *kB = nullptr, *vA = nullptr, *vB = nullptr;
    auto drop = [&]() { dev_free(kA); dev_free(kB); dev_free(vA); dev_free(vB); };
    if ((rc = dev_alloc(&kA, (size_t)n + 256)) || (rc = dev_alloc(&kB, (size_t)n + 256)) || (rc = dev_alloc(&vA, (size_t)n + 256)) ||
        (rc = dev_alloc(&vB, (size_t)n + 256))) { drop(); return rc; }
    hipStream_t s = sl.stream;
    int key_bits = 1;
    while (key_bits < 32 && ((f.key_max - f.key_min) >> key_bits) != 0u) ++key_bits;
    hipLaunchKernelGGL(k_pos_keys, dim3(div_up(n, 256)), dim3(256), 0, s, c->geoA, n, f.cam[0], f.cam[1], f.cam[2], f.key_min, f.key_max, kA, vA);
    rc = radix_sort(sl, kA, vA, kB, vB, n, key_bits, true, (uint32_t*)nullptr, RS_XCD_DEPTH != 0);
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = set_err(GSR_E_HIP, "position-keyed order: sort failed");
    if (rc) { drop(); return rc; }
    dev_free(c->pos_order);
    c->pos_order = vA; vA = nullptr;       // (radix_sort leaves the result in the A pair)
    drop();
    c->pos_valid = true;
    c->pos_gen = c->geo_gen;
    std::memcpy(c->pos_cam, f.cam, sizeof c->pos_cam);
    c->pos_kmin = f.key_min; c->pos_kmax = f.key_max;
    return GSR_OK;
}

static int frame_begin(gsr_context* c, const gsr_camera* cam, const float* depth, int depth_is_device,
                       float* rgba_out, int out_is_device, FrameSlot** used, bool allow_cull, int phase_in /* 2 = the second phase of a front-slab frame */)
{
    if (!c || !cam || !rgba_out) return set_err(GSR_E_INVALID, "gsr_render: NULL argument");
    if (c->uploading) return set_err(GSR_E_INVALID, "gsr_render: upload in progress");
    if (cam->width <= 0 || cam->height <= 0 || cam->width > GSR_MAX_DIM || cam->height > GSR_MAX_DIM)
        return set_err(GSR_E_INVALID, "gsr_render: bad framebuffer size %dx%d (max %d)", cam->width, cam->height, GSR_MAX_DIM);
    if (!c->has_geometry) return set_err(GSR_E_NO_GEOMETRY, "gsr_render: nothing uploaded");
#ifdef GSR_HOST_TIMING
    const double t_enter = now_us();
#endif
    HIP_TRY(hipSetDevice(c->device));

    // Frames alternate between the slots.  Everything up to the blend kernel touches only the slot's
    // private arrays (plus the read-only geometry), so it may run while the previous frame -- and
    // whatever the caller queued on the public stream -- is still executing.
    FrameSlot& sl = c->slot[c->frame_no % (uint64_t)c->nslots];
    if (used) *used = &sl;
    int rc = frame_finish(c, sl);   // a deferred frame of this slot: look at its pair count now
    if (rc) return rc;
    // strictly serial frames run on the public stream itself: nothing to hand over, no events; with two frames in flight a
    // slot runs on its own stream and the result is ordered onto the public stream by an event
    hipStream_t want = (c->nslots == 1) ? c->stream : sl.own;
    if (want != sl.stream) {
        HIP_TRY(hipStreamSynchronize(sl.stream));
        sl.stream = want;
    }
    hipStream_t s = sl.stream;
    if (sl.pair_cap > 0 && sl.pair_cap < c->pair_want) {   // the other slot met a frame that outgrew this size
        HIP_TRY(hipStreamSynchronize(s));
        dev_free(sl.pvA);
        sl.pair_cap = 0;
        if ((rc = dev_alloc(&sl.pvA, c->pair_want + 4))) return rc;
        sl.pair_cap = c->pair_want;
    }

    FrameJob& j = sl.job;
    j = FrameJob();
    build_frame(c, cam, &j.f);
    const GsrFrame& f = j.f;
    const uint32_t n = c->n;
    j.n = n;
    j.local_tiles = f.tiles_x * f.local_tiles_y;
    j.band_rows = (c->shard_count > 1) ? gsr_band_rows(cam->height, c->shard_index, c->shard_count) : cam->height;
    j.out_px = (size_t)j.band_rows * cam->width;
    j.n_super = f.stiles_x * f.stiles_y;
    j.timing = c->opt_timing != 0 && (c->opt_timing >= 2 || c->frame_no % (uint64_t)c->opt_timing_every == 0);
    j.timing_all = c->opt_timing >= 2;   // level 1 brackets only the blend kernel (events 5 and 6)
    j.use_map = c->opt_swizzle != 0;
    j.user_out = rgba_out;
    j.out_is_device = out_is_device != 0;
    j.cam_arg = *cam; j.depth_arg = depth; j.depth_is_device_arg = depth_is_device; j.out_arg = rgba_out; j.out_is_device_arg = out_is_device;
    j.deferred = c->opt_deferred && j.out_is_device && phase_in == 0;
    // order 0: the colour is Cd itself, nothing to defer; mode 1 follows the kernels' own verdict on the previous frames
    j.lazy = f.sh_order > 0 && (c->opt_lazy == 2 || (c->opt_lazy == 1 && c->lazy_pays));
    {
        const int sig[7] = {f.width, f.height, f.shard_index, f.shard_count, f.shard_rpb, f.super_shift, (int)c->geo_gen};
        j.cull = allow_cull && phase_in == 0 && c->opt_cull && !j.deferred && n > 0 && sl.horizon_valid && std::memcmp(sig, sl.horizon_sig, sizeof sig) == 0 &&
                 !(c->opt_flags & GSR_FLAG_FULL_KEYS) && c->opt_cull != 3 && c->cull_pol.allows(c->opt_cull);
        if (j.cull && c->opt_cull == 1 && camera_jumped(c, sl.horizon_cam, *cam)) { j.cull = false; c->st.frames_jumped += 1; }
        if (allow_cull && phase_in == 0) c->cull_pol.tick();
        j.f.cull_dilate = std::max(c->cull_pol.dilate - sl.hpyr_re, 0);   // (the rest of the radius is built into the slot's pyramid)
        if (j.cull) {
            c->st.frames_culled += 1;
            // What K1 keeps is about what the frame composites: it evaluates the colours itself (eager: the SoA colour chunks of a
            // cluster's 64 consecutive splats are coalesced reads) -- unless the frame keeps far more than its tiles look at
            // (oblique ground, silhouettes: then list prefixes + fallback), or lazy colour is forced.  (Round 2 ran a separate pass
            // over the sorted payloads, k_colour_kept: 16 us of scattered 128-byte rows against 9 us more in K1.)
            if (f.sh_order > 0 && c->opt_lazy) j.lazy = c->opt_lazy >= 2 || (c->prefix_cheaper && c->prefix_valid);
        }
    }
    // Front-slab frame?  A frame that cannot be culled against a previous frame's horizons (the first frames of a cloud, a jump, the
    // re-render of a frame that broke a horizon, culling held off) is rendered in two phases where occlusion culling is known to pay:
    // the nearest splats first, then -- behind the tiles that are still open only -- the rest (GSR_OPT_FRONT_SLAB; k_blend.h).
    const SortKey key_now = {c->geo_gen, c->shard_index, c->shard_count, c->shard_layout, c->opt_flags, *cam};
    {
        const bool hit = c->opt_sort_cache && sl.sort_valid && sl.sort_key.same(key_now) && !j.cull && !sl.sorted_culled && !sl.sorted_dculled;   // (a static redraw)
        const bool slab = phase_in == 0 && !j.cull && !j.deferred && n > 0 && c->opt_slab && c->opt_cull && c->opt_cluster && c->bbox_ok &&
                          !(c->opt_flags & GSR_FLAG_FULL_KEYS) && c->opt_sort_cache < 2 && !hit &&
                          // (where a frame is heavy enough for two phases' worth of launches to be repaid: C3's 0.8 M visible splats are not.
                          //  Not tied to cull_weak: horizons of another view cull weakly whatever the scene; a slab that is weak holds ITSELF off)
                          c->slab_pol.allows(c->opt_slab >= 2 || c->opt_cull == 3, c->cull_pol);
        if (phase_in == 0) c->slab_pol.tick();
        j.phase = phase_in == 2 ? 2 : (slab ? 1 : 0);
        j.f.phase = j.phase;
        if (j.phase) {   // (a phase keeps about what it composites: K1 shades on the spot; events only around phase 1's blend kernel)
            j.lazy = false;
            j.timing = j.phase == 1 && j.timing && c->opt_timing == 1;
            j.timing_all = false;
        }
        if (j.phase == 2) j.f.cull_dilate = 0;                 // (this frame's own tiles: nothing moves)
        if (j.phase == 1) c->st.frames_slab += 1;
    }
    if (j.lazy && n > 0 && phase_in == 0) c->st.frames_lazy += 1;
    j.ticket = ++sl.ticket ? sl.ticket : ++sl.ticket;   // (never 0: the mailbox starts at 0)
    if (j.timing) harvest_slot(c, sl);

    // the caller's stream position now: the blend kernel (the only writer of caller-visible memory)
    // waits for it, so an output buffer that earlier work on the public stream still reads is safe
    j.direct = sl.stream == c->stream;
    if (!j.direct) HIP_TRY(hipEventRecord(sl.ev_user, c->stream));

    // per-tile bookkeeping + super-tile ranges
    if ((size_t)j.local_tiles + 1 > sl.tile_cap || !sl.sstart) {
        HIP_TRY(hipStreamSynchronize(s));
        dev_free(sl.tile_work); dev_free(sl.tile_work_a); dev_free(sl.sstart); dev_free(sl.send); dev_free(sl.redo);
        sl.tile_cap = 0;
        if ((rc = dev_alloc(&sl.tile_work, (size_t)j.local_tiles + 1)) || (rc = dev_alloc(&sl.tile_work_a, (size_t)j.local_tiles + 1)) || (rc = dev_alloc(&sl.sstart, (size_t)65536 + 1)) ||
            (rc = dev_alloc(&sl.send, (size_t)65536 + 1)) || (rc = dev_alloc(&sl.redo, (size_t)j.local_tiles + 1))) return rc;
        sl.tile_cap = (size_t)j.local_tiles + 1;
    }
    if (j.phase == 1 && j.out_px > sl.tbuf_cap) {          // the transmittance a front slab leaves behind, per pixel of the band
        HIP_TRY(hipStreamSynchronize(s));
        dev_free(sl.tbuf);
        sl.tbuf_cap = 0;
        if ((rc = dev_alloc(&sl.tbuf, j.out_px))) return rc;
        sl.tbuf_cap = j.out_px;
    }
    if (j.use_map && (rc = build_tile_map(c, f))) return rc;
    j.d_depth = depth;
    if (depth && !depth_is_device) {
        const size_t npx = (size_t)cam->width * cam->height;
        if (npx > sl.depth_cap) {
            HIP_TRY(hipStreamSynchronize(s));
            dev_free(sl.depth_stage);
            sl.depth_cap = 0;
            if ((rc = dev_alloc(&sl.depth_stage, npx))) return rc;
            sl.depth_cap = npx;
        }
        HIP_TRY(hipMemcpyAsync(sl.depth_stage, depth, npx * 4, hipMemcpyHostToDevice, s));
        j.d_depth = sl.depth_stage;
    }
    // Depth-tested frames: the tile-max pyramid of the opaque pass's depth (k_cluster.h), rebuilt every frame (the buffer's content is
    // the caller's), one per slot; the second phase of a front-slab frame uses the first one's.  GSR_OPT_OCCLUSION_CULL = 0 switches
    // it off like every other occlusion test (what is left is k_blend's own per-quadrant classification and per-fragment compare).
    j.dcull = j.d_depth != nullptr && c->opt_cull != 0 && n > 0;
    if (j.dcull) frame_depth_codes(c, cam, &j.f);
    if (j.dcull) {
        const size_t need = (size_t)f.pyr_off[GSR_PYR_LEVELS - 1] + (size_t)gsr_pyr_dim(f.tiles_x, GSR_PYR_LEVELS - 1) * gsr_pyr_dim(f.tiles_y, GSR_PYR_LEVELS - 1) + 16;
        if (need > sl.dpyr_cap || !sl.dactive) {
            HIP_TRY(hipStreamSynchronize(s));
            dev_free(sl.dpyr);
            sl.dpyr_cap = 0;
            if ((rc = dev_alloc(&sl.dpyr, 5 * need))) return rc;                   // (+ the per-tile "covered" array k_blend reads)
            HIP_TRY(hipMemsetAsync(sl.dpyr, 0, 5 * need * sizeof(float), s));   // (levels 4 and 5 start cleared; afterwards every frame clears the next one's)
            sl.dpyr_cap = need;
            if (!sl.dactive) {
                if ((rc = dev_alloc(&sl.dactive, 2))) return rc;
                HIP_TRY(hipMemsetAsync(sl.dactive, 0, 2 * sizeof(uint32_t), s));   // (on the frame's stream: a null-stream memset is not ordered against it)
            }
        }
        if (phase_in != 2) sl.dpar ^= 1;
        j.dpar = sl.dpar;
    }
    j.target = rgba_out;
    if (!out_is_device) {
        if (j.out_px * 4 > sl.fb_cap) {
            HIP_TRY(hipStreamSynchronize(s));
            dev_free(sl.fb);
            sl.fb_cap = 0;
            if ((rc = dev_alloc(&sl.fb, j.out_px * 4))) return rc;
            sl.fb_cap = j.out_px * 4;
            std::memset(sl.fb_sig, 0xff, sizeof sl.fb_sig);
        }
        // A sharded context's band image is padded (gsr_band_rows): the pixel rows behind the rank's last image row are never
        // written.  In the library's own staging buffer they read as zeros: it is cleared whenever the band's shape changes.
        const int fsig[5] = {f.width, f.height, f.shard_index, f.shard_count, f.shard_rpb};
        if (std::memcmp(fsig, sl.fb_sig, sizeof fsig) != 0) {
            HIP_TRY(hipMemsetAsync(sl.fb, 0, j.out_px * 16, s));
            std::memcpy(sl.fb_sig, fsig, sizeof fsig);
        }
        j.target = sl.fb;
    }

    j.open = true;   // from here on kernels are queued: every error path drains them (frame_abort)
    if ((rc = mark(sl, 0))) return frame_abort(sl, rc);
    // The depth order depends on the camera POSITION only (argsortByDistance re-sorts when the position moves,
    // src/GSplatRenderer.C:165-186) -- but the sorted list holds just the splats visible to the frame that sorted, so
    // it is reused as is only for an identical frame description (a static viewport redraw), per frame slot.
    // (a culled frame's order holds only the splats in front of ITS horizons, and the horizons move: no reuse either way; the
    //  two phases of a front-slab frame each sort what they keep)
    // (nor the order of a frame that was culled against its depth buffer: the buffer's CONTENT is not part of the frame description)
    const bool cache_hit = c->opt_sort_cache && sl.sort_valid && sl.sort_key.same(key_now) && !j.cull && !sl.sorted_culled && !sl.sorted_dculled && j.phase == 0;
    // Position-keyed order (GSR_OPT_SORT_CACHE = 2; the reference's rule, src/GSplatRenderer.C:165-186): while the camera POSITION stands
    // still -- a rotation about the eye, a change of lens -- the depth order of ALL splats is the one sorted when it last moved, K1 walks
    // the splats in that order, and what a frame keeps leaves it sorted.  Built the second time a position is seen (a moving camera never
    // pays for it).  Not for sharded, deferred or full-key frames.
    bool ordered = false;
    if (c->opt_sort_cache >= 2 && !cache_hit && n > 0 && c->shard_count == 1 && !j.deferred && !(c->opt_flags & GSR_FLAG_FULL_KEYS) && j.phase == 0) {
        const bool same_pos = c->last_cam_set && std::memcmp(c->last_cam, cam->cam_pos, sizeof c->last_cam) == 0;
        if (c->pos_valid && (c->pos_gen != c->geo_gen || std::memcmp(c->pos_cam, cam->cam_pos, sizeof c->pos_cam) != 0 ||
                             c->pos_kmin != f.key_min || c->pos_kmax != f.key_max))
            c->pos_valid = false;
        if (!c->pos_valid && same_pos) {
            if ((rc = build_pos_order(c, sl, f))) return frame_abort(sl, rc);
        }
        ordered = c->pos_valid;
    }
    std::memcpy(c->last_cam, cam->cam_pos, sizeof c->last_cam);
    c->last_cam_set = true;
    // which depth sort: A frame that keeps few splats (occlusion culling; small clouds) is sorted by ONE bucket scatter + one local
    // kernel (k_sort.h) instead of three global passes: 2 launches instead of 9.  Chosen from what the slot's previous frame kept
    // (correct whatever it chooses).  Not for deferred frames: nobody could render them again; and no prediction from a culled
    // frame for an unculled one.
    int key_bits = 1;
    while (key_bits < 32 && ((f.key_max - f.key_min) >> key_bits) != 0u) ++key_bits;
    const uint32_t n_slots = n ? div_up(c->nclus, 4u) * (uint32_t)GSR_K1_THREADS : 0u;   // the slots K1 can fill at most
    // (back-off: a geometry with a long run of coincident splats fails the small-frame sort's tie rule every frame; the re-render
    //  refills the hints, so without this it would be rendered twice per frame for good)
    const bool held = sl.local_pol.begin_frame(c->opt_local_sort, cache_hit);
    const bool local = !cache_hit && !ordered && n_slots > 0 && key_bits > 9 && !(c->opt_flags & GSR_FLAG_FULL_KEYS) && sl.kept_hi > sl.kept_lo && !j.deferred &&
                       !c->classic_once && !held && sl.kept_culled == j.cull && j.phase == 0 &&
                       (c->opt_local_sort >= 2 || (c->opt_local_sort == 1 && sl.kept_hint > 0 && sl.kept_hint <= 500000u));
    const bool classic_now = c->classic_once;
    if (!cache_hit && j.phase != 1) c->classic_once = false;   // (the re-render of a front-slab frame: both phases)
    // front-slab phases: the key range of a phase is k_slab_pick's (on the device); the small-frame sort is taken when the slot's
    // last front-slab frame kept few enough in that phase (phase 1: the slab holds <= slab_max clusters; the first such frame: global passes)
    const uint32_t kept_prev = j.phase == 1 ? sl.slab_kept1 : sl.slab_kept2;
    const bool local_phase = j.phase != 0 && n_slots > 0 && key_bits > 9 && c->opt_local_sort && !classic_now && !held && kept_prev > 0 && kept_prev <= 900000u;
    j.local_sort = local || local_phase;
    GsrK1Scatter scat{};
    uint32_t bk_lo = 0u;
    int bk_shift = 0;
    bool k1_scatters = false;
    if (n > 0) {
#ifdef GSR_HOST_TIMING
        const double t_pre = now_us();
#endif
        // cluster culling (k_cluster.h): which clusters of 64 storage-ordered splats can draw anything in this frame
        int rounds; uint32_t ngroups;
        cluster_grid(c->nclus, &rounds, &ngroups);
        if ((c->opt_flags & GSR_FLAG_CULL_ROUNDS) && c->nclus > 0) {   // (test hook: the several-rounds-per-workgroup form clouds beyond 33 M splats take)
            rounds = std::max(rounds, 3);
            ngroups = div_up(c->nclus, (uint32_t)CC_THREADS * (uint32_t)rounds);
        }
        // (front-slab frames: phase 1 leaves a histogram of the survivors' nearest keys and k_slab_pick takes the slab key from it;
        //  phase 2 culls against the tiles phase 1 finished)
        const int hist_shift = key_bits > 10 ? key_bits - 10 : 0;
        const float* pyr = j.phase == 2 ? sl.hpyr2 : ((j.cull && !ordered) ? sl.hpyr : (const float*)nullptr);
        const GsrSlabPick pk{(uint32_t)c->slab_min, (uint32_t)c->slab_max, (uint32_t)c->slab_frac, f.key_max - f.key_min};
        // depth-tested frames: the opaque pass's tile-max depth pyramid -- where the previous depth-tested frame found opaque geometry in
        // its buffer, in a launch of its own in front of everything (k_cluster_cull and K1 cull against it); otherwise -- a buffer
        // cleared to the far plane, the common case -- NO pyramid: the first workgroups of k_cluster_cull's launch only look whether a
        // pixel is covered at all (k_cluster.h: gsr_depth_detect_block), and a frame in which geometry appears goes without depth culling.
        // Whatever the guess, the pixels are the same: the tests are conservative and k_blend compares every fragment.
        GsrDepthPyrArgs dp{};
        uint32_t n_dp = 0;
        GsrDepthCull dc_clus{nullptr, nullptr, nullptr}, dc_k1{nullptr, nullptr, nullptr};
        if (j.dcull) {
            float* const dp0 = sl.dpyr + (size_t)(2 * j.dpar) * sl.dpyr_cap;
            float* const dq0 = sl.dpyr + (size_t)(2 * (j.dpar ^ 1)) * sl.dpyr_cap;
            dc_k1 = GsrDepthCull{dp0, j.phase == 2 ? (const float*)nullptr : dp0 + sl.dpyr_cap, sl.dactive + j.dpar};
            if (j.phase == 2 && !sl.dpyr_built) { dc_k1 = GsrDepthCull{nullptr, nullptr, nullptr}; j.dblind = true; }   // (phase 1 built none)
            if (j.phase != 2) {
                dp.depth = j.d_depth; dp.pyr = dp0; dp.pyrc = dp0 + sl.dpyr_cap; dp.pyr_next = dq0; dp.pyrc_next = dq0 + sl.dpyr_cap; dp.active = sl.dactive; dp.par = j.dpar;
                dp.tcov = sl.dpyr + 4 * sl.dpyr_cap;
                // (a culled frame: the tiles that were classic when the horizons were left get no depth clause)
                dp.stat = (j.cull && sl.hstat_valid) ? sl.hstat : (const float*)nullptr;
                j.dstat = dp.stat != nullptr;
                dp.width = f.width; dp.height = f.height; dp.tiles_x = f.tiles_x; dp.tiles_y = f.tiles_y;
                for (int l = 0; l < GSR_PYR_LEVELS; ++l) dp.off[l] = f.pyr_off[l];
                const uint32_t nb8 = (uint32_t)gsr_depth_pyramid_blocks(f.tiles_x, f.tiles_y);
                if (c->depth_active) {
                    // (measured and rejected, round 6: the pass inside k_cluster_cull's launch here too, the cluster workgroups waiting on a
                    //  count of finished pyramid workgroups before their depth tests: 3650 fps against 4245 -- LAB_NOTES.md)
                    hipLaunchKernelGGL(k_depth_pyramid, dim3(nb8), dim3(1024), 0, s, dp);
                    if (c->opt_cluster && !ordered) dc_clus = dc_k1;
                    sl.dpyr_built = true;
                } else {
                    (void)nb8;
                    n_dp = (uint32_t)gsr_depth_detect_blocks(f.width, f.height);
                    j.dblind = true;
                    j.dstat = false;
                    sl.dpyr_built = false;
                    dc_k1 = GsrDepthCull{nullptr, nullptr, nullptr};
                }
            } else if (c->opt_cluster && sl.dpyr_built) {
                dc_clus = dc_k1;       // (phase 1 built it)
            }
        }
        if (j.phase == 1) {   // a first pass for the histogram alone (the pass below takes the slab key from it and keeps the slab's clusters only)
            // (the histogram is cleared by phase 2's cull pass; a phase 1 that never got its phase 2 -- its small-frame sort gave a bucket
            //  up, an error in between -- left its counts behind: they would skew this frame's slab key, never its pixels)
            if (sl.slab_dirty && hipMemsetAsync(sl.slab, 0, (size_t)GSR_SLAB_BINS * sizeof(uint32_t), s) != hipSuccess)
                return frame_abort(sl, set_err(GSR_E_HIP, "gsr_render: clearing the slab histogram failed"));
            sl.slab_dirty = true;
            const int n45 = gsr_pyr_dim(f.tiles_x, 4) * gsr_pyr_dim(f.tiles_y, 4) + gsr_pyr_dim(f.tiles_x, 5) * gsr_pyr_dim(f.tiles_y, 5);
            hipLaunchKernelGGL(j.dcull ? k_cluster_cull<true> : k_cluster_cull<false>, dim3(ngroups + n_dp), dim3(CC_THREADS), 0, s, f, c->clusA, c->clusB, c->nclus, rounds, c->opt_cluster,
                               (const float*)nullptr, sl.cseg, sl.ccnt, 1, sl.slab, hist_shift, sl.slab + GSR_SLAB_BINS, pk, sl.hpyr2 + f.pyr_off[4], n45,
                               (uint32_t*)nullptr, (uint32_t*)nullptr, dp, n_dp, dc_clus);
            n_dp = 0;      // (looked at: not again in the pass below)
        }
        if (j.phase == 2) sl.slab_dirty = false;   // (mode 3 below clears the histogram for the slot's next front-slab frame)
        hipLaunchKernelGGL(j.dcull ? k_cluster_cull<true> : k_cluster_cull<false>, dim3(ngroups + n_dp), dim3(CC_THREADS), 0, s, f, c->clusA, c->clusB, c->nclus, rounds, ordered ? 0 : c->opt_cluster,
                           pyr, sl.cseg, sl.ccnt,   // (ordered: slots, not clusters -- all of them)
                           j.phase == 1 ? 2 : (j.phase == 2 ? 3 : 0), sl.slab, hist_shift, sl.slab + GSR_SLAB_BINS, pk, (float*)nullptr, 0,
                           (local || local_phase) ? sl.bkt_cnt : (uint32_t*)nullptr, sl.d_counts + 2, dp, n_dp, dc_clus);
        // The small-frame sort's bucket pass runs inside K1 (a key's place in its bucket = one atomic): BK_BUCKETS buckets of equal width
        // over the key range the previous frame kept, widened by a sixteenth on either side (the view moves), in this frame's key domain
        // (keys are stored relative to key_min); front-slab phases: over the phase's own range, which k_slab_pick left on the device
        if (local) {
            const uint64_t span = (uint64_t)sl.kept_hi - sl.kept_lo, margin = span / 16 + 64;
            const uint64_t lo_abs = sl.kept_lo > margin ? sl.kept_lo - margin : 0, hi_abs = (uint64_t)sl.kept_hi + margin;
            bk_lo = lo_abs > f.key_min ? (uint32_t)(lo_abs - f.key_min) : 0u;
            const uint64_t width = (hi_abs > f.key_min ? hi_abs - f.key_min : 0) - bk_lo + 1;
            while (bk_shift < 31 && (width >> bk_shift) > (uint64_t)BK_BUCKETS) ++bk_shift;
        }
        k1_scatters = c->opt_k1_scatter != 0 && (local || local_phase) && c->opt_scatter_direct >= 0;
        if (k1_scatters) {
            scat.key = sl.bkt_key; scat.val = sl.bkt_val; scat.cnt = sl.bkt_cnt; scat.failed = sl.d_counts + 2;
            scat.range_dev = local_phase ? sl.slab + GSR_SLAB_BINS + (j.phase == 1 ? 2 : 4) : (const uint32_t*)nullptr;
            scat.lo = bk_lo; scat.shift = bk_shift;
        }
        // K1 over the survivors, four clusters per workgroup-iteration; the grid follows the slot's previous frame (+25 %), and
        // a frame that keeps more simply loops
        const uint32_t all_iter = div_up(c->nclus, 4u);
        uint32_t k1_grid = all_iter;
        if (sl.surv_hint > 0 && !ordered) k1_grid = std::min<uint32_t>(all_iter, div_up(sl.surv_hint, 4u) * 5u / 4u + 64u);
        if (j.phase) k1_grid = std::min<uint32_t>(all_iter, 8192u);   // (what a phase keeps is not known beforehand: a bounded grid that loops)
        // on a cache hit (identical frame description) the sorted (keyA, valA) are kept and K1's key/payload
        // output goes to the scratch buffers
        j.k1_grid = k1_grid;
        // (two instantiations: the one that leaves the colours pending has no SH evaluation in it and runs at 8 waves per SIMD instead of 6)
        hipLaunchKernelGGL(j.d_depth ? (j.lazy ? k_preprocess_lazy_depth : k_preprocess_depth) : (j.lazy ? k_preprocess_lazy : k_preprocess), dim3(k1_grid ? k1_grid : 1u), dim3(GSR_K1_THREADS), 0, s, n, c->cap, f, c->geoA, c->geoB, c->col,
                           sl.rec, (cache_hit || ordered) ? sl.keyB : sl.keyA, (cache_hit || ordered) ? sl.valB : sl.valA,
                           j.d_depth ? sl.zwin : (float*)nullptr, j.phase == 2 ? sl.hpyr2 : (j.cull ? sl.hpyr : (const float*)nullptr), sl.blk_cnt,
                           sl.cseg, sl.ccnt, ngroups, (uint32_t)CC_THREADS * (uint32_t)rounds, sl.d_counts, scat,
                           // (the count of sorted splats starts at zero: a frame whose clusters are ALL culled runs no sort workgroup that
                           //  could say so, and the binning kernels would walk the previous frame's order; a static redraw keeps its order)
                           cache_hit ? (uint32_t*)nullptr : sl.d_n,
                           ordered ? c->pos_order : (const uint32_t*)nullptr, sl.slab + GSR_SLAB_BINS, dc_k1);
        hipError_t e = hipGetLastError();
#ifdef GSR_HOST_TIMING
        if (g_t_verdict > 0) {
            const double t_l = now_us();
            g_acc_py += t_enter - g_t_verdict; g_acc_pre += t_pre - t_enter; g_acc_launch += t_l - t_pre; ++g_acc_n;
            if (g_acc_n % 100 == 0)
                fprintf(stderr, "[host timing] verdict -> gsr_render %.1f us, frame_begin before K1 %.1f us, K1 launch %.1f us (avg of %ld)\n",
                        g_acc_py / g_acc_n, g_acc_pre / g_acc_n, g_acc_launch / g_acc_n, g_acc_n);
            g_t_verdict = 0;
        }
#endif
        if (e != hipSuccess) return frame_abort(sl, set_err(GSR_E_HIP, "k_preprocess: %s", hipGetErrorString(e)));
    }
    if ((rc = mark(sl, 1))) return frame_abort(sl, rc);
    if (cache_hit) {
        c->st.sorts_skipped += 1;
    } else if (ordered) {
        // K1 walked the splats nearest first: the heads of its 256-slot blocks, one after the other, ARE the sorted frame
        const uint32_t m_max = n_slots / RS_SRC_BLOCK;
        if ((rc = ensure_u32(&c->blk_pre, &c->blk_pre_cap, (size_t)m_max + 8))) return frame_abort(sl, rc);
        hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, sl.blk_cnt, sl.d_counts, c->blk_pre, sl.d_n);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_compact_blocks<uint2>), dim3(m_max), dim3(RS_SRC_BLOCK), 0, s, sl.keyB, sl.valB, sl.blk_cnt, c->blk_pre,
                           sl.d_counts, sl.keyA, sl.valA);
        if (hipGetLastError() != hipSuccess) return frame_abort(sl, set_err(GSR_E_HIP, "position-keyed order: launch failed"));
        c->st.sorts_skipped += 1;
        sl.key_min = f.key_min;
        sl.sort_valid = false;       // (what the slot holds is this frame's kept set only)
        sl.sorted_culled = j.cull;
    } else {
        if (local_phase) {
            // K1's compacted slots -> bucket regions over the phase's key range (k_slab_pick left it on the device) -> sorted (keyA, valA)
            const uint32_t* range = sl.slab + GSR_SLAB_BINS + (j.phase == 1 ? 2 : 4);
            if (!k1_scatters)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bucket_scatter_k1<uint2>), dim3(std::max(1u, std::min(div_up(n_slots / RS_SRC_BLOCK, 4u), div_up(j.k1_grid, 4u) + 16u))), dim3(256), 0, s, sl.keyA, sl.valA,
                                   n_slots / RS_SRC_BLOCK, sl.d_counts, 0, 0u, sl.blk_cnt, sl.bkt_cnt, sl.bkt_key, sl.bkt_val, sl.d_counts + 2, range);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_radix_local<uint2>), dim3(BK_BUCKETS), dim3(RL_THREADS), 0, s, sl.bkt_cnt, 0, key_bits, 0u,
                               sl.bkt_key, sl.bkt_val, sl.keyA, sl.valA, sl.d_counts + 2, sl.d_n, range);
            if (hipGetLastError() != hipSuccess) rc = set_err(GSR_E_HIP, "small-frame sort: launch failed");
        } else if (local) {
            // BK_BUCKETS buckets of equal width over the key range the previous frame kept, widened by a sixteenth on either side (the
            // view moves); in this frame's key domain (keys are stored relative to key_min)
            const uint32_t lo = bk_lo;
            const int bshift = bk_shift;
            const uint32_t nblk = div_up(n_slots, RS_TILE);
            // K1's compacted slots -> bucket regions (counters and *d_n were cleared by K1) -> sorted (keyA, valA)
            // one global atomic per key pays up to ~150 k keys (fps direct / aggregated: C1 16 700 / 14 400, C2 8830 / 8560, C3 4650 / 4920, C4 3800 / 3950)
            const bool direct = c->opt_scatter_direct == 2 || (c->opt_scatter_direct == 1 && sl.kept_hint <= 150000u);
            if (k1_scatters) {
                // (K1 did it)
            } else if (direct)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bucket_scatter_direct<uint2>), dim3(n_slots / RS_SRC_BLOCK), dim3(RS_SRC_BLOCK), 0, s, sl.keyA, sl.valA,
                                   sl.d_counts, bshift, lo, sl.blk_cnt, sl.bkt_cnt, sl.bkt_key, sl.bkt_val, sl.d_counts + 2);
            else if (c->opt_scatter_direct >= 0)
                // (grid: K1's own estimate of its workgroup-iterations, four per workgroup here)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bucket_scatter_k1<uint2>), dim3(std::max(1u, std::min(div_up(n_slots / RS_SRC_BLOCK, 4u), div_up(j.k1_grid, 4u) + 16u))), dim3(256), 0, s, sl.keyA, sl.valA,
                                   n_slots / RS_SRC_BLOCK, sl.d_counts, bshift, lo, sl.blk_cnt, sl.bkt_cnt, sl.bkt_key, sl.bkt_val, sl.d_counts + 2);
            else   // (A/B: the general gathering scatter, 2048 slots per workgroup)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bucket_scatter<uint2, true>), dim3(nblk), dim3(RS_THREADS), 0, s, sl.keyA, sl.valA, n_slots, sl.d_counts,
                                   bshift, lo, sl.blk_cnt, sl.bkt_cnt, sl.bkt_key, sl.bkt_val, (uint32_t*)nullptr, sl.d_counts + 2);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_radix_local<uint2>), dim3(BK_BUCKETS), dim3(RL_THREADS), 0, s, sl.bkt_cnt, bshift, key_bits, lo,
                               sl.bkt_key, sl.bkt_val, sl.keyA, sl.valA, sl.d_counts + 2, sl.d_n);
            if (hipGetLastError() != hipSuccess) rc = set_err(GSR_E_HIP, "small-frame sort: launch failed");
        } else {
            // (keys per thread: by what the slot's previous frame kept -- any choice sorts correctly)
            // (... from a frame of the same kind: an unculled frame keeps ten times what a culled one does)
            const bool mid = c->opt_mid_sort && sl.kept_hint > 0 && sl.kept_hint <= 1500000u && n_slots <= 4000000u && sl.kept_culled == j.cull;
            rc = radix_sort(sl, sl.keyA, sl.valA, sl.keyB, sl.valB, n_slots, key_bits,
                            !(c->opt_flags & GSR_FLAG_FULL_KEYS), sl.d_n, RS_XCD_DEPTH != 0, sl.blk_cnt, sl.d_counts, mid);
        }
        if (rc) return frame_abort(sl, rc);
        sl.key_min = f.key_min;
        sl.sort_valid = j.phase == 0;    // (a phase's order holds a part of the frame only)
        sl.sort_key = key_now;
        sl.sorted_culled = j.cull;
        sl.sorted_dculled = j.dcull;     // (until the frame's mailbox says that its depth buffer culled nothing: frame_finish)
        j.sort_fresh = true;
    }
    if ((rc = mark(sl, 2))) return frame_abort(sl, rc);
    hipError_t e = hipSuccess;
    if (n > 0) {
        // coarse binning as a counting sort (k_binning.h): count -> scan -> ranges -> [pair count to the host] -> place
        // splats per binning workgroup: 1024 for frames that keep millions, fewer for the small ones (k_binning.h)
        // (measured, fps with 4 / 2 / 1: C1 12 770 / 13 540 / 14 190, C2 7880 / 8270 / 8510, C3 4770 / 4900 / 4830, C4 3930 / 3950 / 3760)
        j.bn_items = c->opt_bn_items > 0 ? c->opt_bn_items : (j.phase ? 2 : (!local ? 4 : (sl.kept_hint <= 150000u ? 1 : 2)));
        const uint32_t bn_tile = (uint32_t)BN_THREADS * (uint32_t)j.bn_items;
        const uint32_t nblk = div_up(n, bn_tile);
        rc = ensure_u32(&sl.hist, &sl.hist_cap, (size_t)BN_BINS * nblk + 8);
        if (rc) return frame_abort(sl, rc);
        const GsrShard shd{f.shard_index, f.shard_count, f.shard_rpb, f.rect_shift};
        // the grids of the binning kernels: what the slot's previous frame kept, + 25 % (they loop if the frame keeps more)
        j.bn_grid = (sl.kept_hint > 0 && j.phase == 0) ? std::min<uint32_t>(nblk, div_up(sl.kept_hint + sl.kept_hint / 4u, bn_tile) + 64u) : (j.phase ? std::min<uint32_t>(nblk, 4096u) : nblk);
        // (+ the extra work items of the blocks that are split by rows of super-tiles: k_binning.h, BN_SPLIT_TILES)
        const uint32_t bn_extra = (uint32_t)BN_SPLIT_TILES * (uint32_t)std::max(f.stiles_y - 1, 0);
#define GSR_COUNT(I) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bin_count<I>), dim3(j.bn_grid + bn_extra), dim3(BN_THREADS), 0, s, sl.valA, sl.d_n, f.super_shift - f.rect_shift, \
                                        shd, f.stiles_x, f.stiles_y, sl.hist, nblk)
        if (j.bn_items == 1) GSR_COUNT(1); else if (j.bn_items == 2) GSR_COUNT(2); else GSR_COUNT(4);
#undef GSR_COUNT
        hipLaunchKernelGGL(k_scan_rows, dim3(BN_BINS), dim3(SC_THREADS), 0, s, sl.hist, nblk, sl.totals, sl.d_n, n, bn_tile);
        // the list ranges and the pair count: formed by k_bin_place itself (queue_back_end) when the back end is queued
        // speculatively; a frame without a list buffer needs the count first
        if (!(sl.pair_cap > 0)) hipLaunchKernelGGL(k_bin_ranges, dim3(1), dim3(BN_BINS), 0, s, range_args(c, sl));
        e = hipGetLastError();
    } else {
        e = hipMemsetAsync(sl.sstart, 0, ((size_t)j.n_super + 1) * 4, s);
        if (e == hipSuccess) e = hipMemsetAsync(sl.send, 0, ((size_t)j.n_super + 1) * 4, s);
    }
    if (e != hipSuccess) return frame_abort(sl, set_err(GSR_E_HIP, "gsr_render: binning: %s", hipGetErrorString(e)));
    j.speculative = n > 0 && sl.pair_cap > 0;
    j.ranges_folded = j.speculative;
    if (j.speculative || n == 0) {
        if ((rc = queue_back_end(c, sl))) return frame_abort(sl, rc);
    } else {
        j.deferred = false;   // no list buffer yet (first frame): the host has to size it before anything is composited
    }
    if (j.deferred && (rc = queue_frame_end(c, sl))) return frame_abort(sl, rc);
    sl.last_supers = j.n_super;
    sl.last_tiles_x = f.tiles_x;
    sl.last_local_ty = f.local_tiles_y;
    sl.super_tile = f.super; sl.stiles_x = f.stiles_x; sl.stiles_y = f.stiles_y;
    c->st.frames += 1;
    c->frame_no += 1;
    sl.frame_id = c->frame_no;
    // (the frame's counters are written to the host mirror by k_sum_work; they are read in gsr_get_stats)
    return GSR_OK;
}

static FrameSlot* latest_slot(gsr_context* c);
// a culled frame is handed over only once it has checked itself; one that broke a horizon is rendered again, complete
static int frame_check(gsr_context* c, FrameSlot& first, const gsr_camera* cam, const float* depth, int depth_is_device,
                       float* rgba_out, int out_is_device)
{
    // A frame that is rendered again is checked again: the repair of a culled frame may itself meet a bucket its small-frame sort
    // gives up, or -- as a front-slab frame -- a list buffer its second phase overruns.  Every reason removes itself (the re-sort
    // takes the global passes, the redo finds the buffer regrown, the repair does not cull), so a handful of rounds is the most a
    // frame can need; found by tools/fuzz_parity.py: the second re-render used to be handed over unchecked.
    FrameSlot* cur = &first;
    for (int round = 0; round < 8; ++round) {
        FrameSlot& slot = *cur;
        if (slot.job.sort_failed || slot.job.redo) {
            // the small-frame sort met a bucket far beyond its prediction and left it unsorted: the frame again, with the global sort
            // (and without culling: a prediction that far off means the view jumped, and the horizons with it) -- or the second phase
            // of a front-slab frame ran off its list buffer (frame_finish): the frame again, as it was
            if (slot.job.sort_failed) {
                c->st.frames_resorted += 1;
                c->classic_once = true;
            }
            c->frame_no -= 1;
            c->st.frames -= 1;
            FrameSlot* sl2 = nullptr;
            int rc2 = frame_begin(c, cam, depth, depth_is_device, rgba_out, out_is_device, &sl2, false);
            if (rc2) return rc2;
            if ((rc2 = frame_finish(c, *sl2))) return rc2;
            cur = sl2;
            continue;
        }
        if (!slot.job.cull) return GSR_OK;
        bool broke = false;
        int rc = frame_verdict(c, slot, &broke);
        if (rc) return rc;
        if (!broke) {
            c->cull_pol.on_frame_held();
#ifdef GSR_HOST_TIMING
            g_t_verdict = now_us();
#endif
            return GSR_OK;
        }
        c->st.frames_repaired += 1;
        if (slot.dbg_viol) {
            uint32_t hv[80];
            (void)hipStreamSynchronize(slot.stream);
            if (hipMemcpy(hv, slot.dbg_viol, sizeof hv, hipMemcpyDeviceToHost) == hipSuccess) {
                fprintf(stderr, "[viol] frame %llu: %u tile(s) broke their promise\n", (unsigned long long)c->frame_no, hv[0]);
                for (uint32_t k = 0; k < std::min(hv[0], 8u); ++k) {
                    const uint32_t* o = hv + 1 + 8 * k;
                    float hold, kl; std::memcpy(&hold, o + 4, 4); std::memcpy(&kl, o + 5, 4);
                    fprintf(stderr, "   tile (%u, %u): w.w %#x (opaque %u, met %u, u_all %u, steps all %u / u %u) entries read %u, checked depth %u, hold %.4f, last key %.4f, predicted classic %u, kept %u, c1 %u, classic now %u\n",
                            o[0], o[1], o[2], o[2] & 1u, (o[2] >> 14) & 1u, (o[2] >> 15) & 1u, (o[2] >> 16) & 0xffu, o[2] >> 24, o[3], o[7], std::sqrt(hold), std::sqrt(kl), o[6] & 1u, (o[6] >> 1) & 1u, (o[6] >> 2) & 1u, (o[6] >> 3) & 1u);
                }
            }
        }
        // The view is changing faster than the horizons follow.  First answer: compare rects with the horizons of a wider
        // neighbourhood from now on (the repaired frame leaves fresh horizons, and the radius shrinks back while frames hold);
        // only when that is exhausted, leave culling alone for a while (8, 32, 128, 512, 1024 frames).
        c->cull_pol.on_horizon_broke();
        c->frame_no -= 1;          // the same frame again, in the same slot
        c->st.frames -= 1;
        FrameSlot* sl = nullptr;
        rc = frame_begin(c, cam, depth, depth_is_device, rgba_out, out_is_device, &sl, false);
        if (rc) return rc;
        if ((rc = frame_finish(c, *sl))) return rc;
        cur = sl;
    }
    return set_err(GSR_E_HIP, "gsr_render: the frame did not settle after eight attempts");
}

extern "C" int gsr_render(gsr_context* c, const gsr_camera* cam, float* rgba_out, int out_is_device)
{
    return gsr_render_depth(c, cam, nullptr, 0, rgba_out, out_is_device);
}

extern "C" int gsr_render_depth(gsr_context* c, const gsr_camera* cam, const float* depth, int depth_is_device,
                                float* rgba_out, int out_is_device)
{
    FrameSlot* sl = nullptr;
#ifdef GSR_HOST_PHASES
    static double acc[4] = {0, 0, 0, 0}, t_last_exit = 0; static long cnt = 0;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
#endif
    int rc = frame_begin(c, cam, depth, depth_is_device, rgba_out, out_is_device, &sl, true);
    if (rc) return rc;
    if (sl->job.deferred) return GSR_OK;   // GSR_OPT_DEFERRED_CHECK: the pair count is looked at by the next call that syncs
#ifdef GSR_HOST_PHASES
    const double t1 = now();
#endif
    if ((rc = frame_finish(c, *sl))) return rc;
#ifdef GSR_HOST_PHASES
    const double t2 = now();
#endif
    rc = frame_check(c, *sl, cam, depth, depth_is_device, rgba_out, out_is_device);
#ifdef GSR_HOST_PHASES
    const double t3 = now();
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; if (t_last_exit > 0) acc[3] += t0 - t_last_exit;
    t_last_exit = t3;
    if (++cnt % 50 == 0) { fprintf(stderr, "[host phases] begin %.1f  finish(pair wait + frame end) %.1f  check(verdict wait) %.1f  outside gsr_render %.1f us\n", acc[0] / 50, acc[1] / 50, acc[2] / 50, acc[3] / 50); acc[0] = acc[1] = acc[2] = acc[3] = 0; }
#endif
    return rc;
}

// split form for callers that drive several contexts from one thread (gsr_multi.cpp)
__attribute__((visibility("hidden"))) int gsr_internal_frame_begin(gsr_context* c, const gsr_camera* cam, const float* depth,
                                                                    int depth_is_device, float* out_dev)
{
    return frame_begin(c, cam, depth, depth_is_device, out_dev, 1, nullptr, true);
}
__attribute__((visibility("hidden"))) int gsr_internal_frame_finish(gsr_context* c) { return c ? finish_open_frames(c) : GSR_OK; }
// third step of the split form: the newest frame's occlusion-culling verdict (and the repair, if it broke a horizon)
__attribute__((visibility("hidden"))) int gsr_internal_frame_check(gsr_context* c, const gsr_camera* cam, const float* depth,
                                                                    int depth_is_device, float* out_dev)
{
    FrameSlot* sl = c ? latest_slot(c) : nullptr;
    return sl ? frame_check(c, *sl, cam, depth, depth_is_device, out_dev, 1) : GSR_OK;
}
__attribute__((visibility("hidden"))) void* gsr_internal_stream(gsr_context* c) { return c ? (void*)c->stream : nullptr; }
__attribute__((visibility("hidden"))) int gsr_internal_device(gsr_context* c) { return c ? c->device : -1; }

// Wireframe overlay: synchronous, not on the per-frame beauty path.
static int render_wire(gsr_context* c, const gsr_camera* cam, float* rgba_out, int out_is_device, int over);
extern "C" int gsr_render_wire(gsr_context* c, const gsr_camera* cam, float* rgba_out, int out_is_device) { return render_wire(c, cam, rgba_out, out_is_device, 0); }
// wire-over display (the reference draws the outlines AND keeps the primitive in the splat pass, src/GR_GSplat.C:471-486): the outlines
// go on top of the frame that is already in rgba_inout; pixels no outline covers are left as they are
extern "C" int gsr_render_wire_over(gsr_context* c, const gsr_camera* cam, float* rgba_inout, int is_device) { return render_wire(c, cam, rgba_inout, is_device, 1); }
static int render_wire(gsr_context* c, const gsr_camera* cam, float* rgba_out, int out_is_device, int over)
{
    if (!c || !cam || !rgba_out) return set_err(GSR_E_INVALID, "gsr_render_wire: NULL argument");
    if (c->uploading) return set_err(GSR_E_INVALID, "gsr_render_wire: upload in progress");
    if (cam->width <= 0 || cam->height <= 0 || cam->width > GSR_MAX_DIM || cam->height > GSR_MAX_DIM)
        return set_err(GSR_E_INVALID, "gsr_render_wire: bad framebuffer size %dx%d", cam->width, cam->height);
    if (!c->has_geometry) return set_err(GSR_E_NO_GEOMETRY, "gsr_render_wire: nothing uploaded");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    hipStream_t s = c->slot[0].own;
    GsrFrame f;
    build_frame(c, cam, &f);
    const size_t npix = (size_t)cam->width * cam->height;
    // the z-buffer (and the staging image for a host target) are kept between calls: an overlay redraw allocates nothing
    if (npix > c->wire_cap) {
        dev_free(c->wire_zbuf); dev_free(c->wire_out);
        c->wire_cap = 0;
        if ((rc = dev_alloc(&c->wire_zbuf, npix)) || (rc = dev_alloc(&c->wire_out, npix * 4))) {
            dev_free(c->wire_zbuf); dev_free(c->wire_out);
            return rc;
        }
        c->wire_cap = npix;
    }
    // outlines at the same depth are drawn in UPLOAD order (k_wire.h): with spatially ordered storage the resolve pass goes back
    // from the winner's upload index to its slot through the inverse of the storage permutation, built once per geometry
    if (c->perm && c->n > 0 && (!c->wire_inv || c->wire_inv_gen != c->geo_gen)) {
        dev_free(c->wire_inv);
        if ((rc = dev_alloc(&c->wire_inv, (size_t)c->n))) return rc;
        hipLaunchKernelGGL(k_invert_perm, dim3(div_up(c->n, 256)), dim3(256), 0, s, c->perm, c->n, c->wire_inv);
        c->wire_inv_gen = c->geo_gen;
    }
    const uint32_t* const inv = (c->perm && c->n > 0) ? c->wire_inv : (const uint32_t*)nullptr;
    unsigned long long* zbuf = c->wire_zbuf;
    float* target = out_is_device ? rgba_out : c->wire_out;
    hipError_t e = hipMemsetAsync(zbuf, 0xff, npix * 8, s);
    if (e == hipSuccess && over && !out_is_device) e = hipMemcpyAsync(c->wire_out, rgba_out, npix * 16, hipMemcpyHostToDevice, s);   // (the frame underneath)
    if (e == hipSuccess && c->n > 0)
        hipLaunchKernelGGL(k_wire_splats, dim3(div_up(c->n, 256)), dim3(256), 0, s, c->n, f, c->geoA, c->geoB, zbuf, inv ? c->perm : (const uint32_t*)nullptr);
    if (e == hipSuccess)
        hipLaunchKernelGGL(k_wire_resolve, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, zbuf, npix, c->col,
                           reinterpret_cast<float4*>(target), inv, over);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && !out_is_device) e = hipMemcpyAsync(rgba_out, c->wire_out, npix * 16, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return set_err(GSR_E_HIP, "gsr_render_wire: %s", hipGetErrorString(e));
    return GSR_OK;
}

// test door of the policies (gsr_policy.h): a pure function (state, event, a, b) -> state; needs no context and no GPU
extern "C" int gsr_debug_policy(int32_t* state16, int event, long long a, long long b)
{
    if (!state16) return set_err(GSR_E_INVALID, "gsr_debug_policy: NULL state");
    gsr_policy_apply(state16, event, a, b);
    return state16[11] < 0 ? set_err(GSR_E_INVALID, "gsr_debug_policy: unknown event %d", event) : GSR_OK;
}
// ... and the live policy state of a context, in the same layout ([12] = frames the slot's local-sort policy is for: slot 0)
extern "C" int gsr_debug_policy_state(gsr_context* c, int32_t* state16)
{
    if (!c || !state16) return set_err(GSR_E_INVALID, "gsr_debug_policy_state: NULL argument");
    const GsrCullPolicy& p = c->cull_pol;
    const int32_t v[GSR_POLICY_STATE_INTS] = {p.pays, p.weak, (int32_t)p.vis_unculled, p.holdoff, p.backoff, p.streak, p.dilate, p.opt_dilate,
                                              c->slab_pol.holdoff, c->slot[0].local_pol.fails, c->slot[0].local_pol.holdoff, 0, 0, 0, 0, 0};
    std::memcpy(state16, v, sizeof v);
    return GSR_OK;
}

extern "C" int gsr_synchronize(gsr_context* c)
{
    if (!c) return set_err(GSR_E_INVALID, "gsr_synchronize: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    return rc ? rc : gsr_internal_comm_sync(c);   // (... and the gather in flight, if the context has a communicator)
}

// the frame summaries live in device memory (no PCIe writes on the per-frame path); callers have synchronised
static void fetch_counters(gsr_context* c)
{
    for (int k = 0; k < GSR_MAX_SLOTS; ++k)
        if (c->slot[k].d_frame && c->slot[k].h_counters)
            (void)hipMemcpy(c->slot[k].h_counters, c->slot[k].d_frame, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}

// the slot that rendered the most recent frame (NULL before the first frame)
static FrameSlot* latest_slot(gsr_context* c)
{
    FrameSlot* best = nullptr;
    for (int k = 0; k < GSR_MAX_SLOTS; ++k)
        if (c->slot[k].frame_id && (!best || c->slot[k].frame_id > best->frame_id)) best = &c->slot[k];
    return best;
}

extern "C" int gsr_get_stats(gsr_context* c, gsr_stats* out)
{
    if (!c || !out) return set_err(GSR_E_INVALID, "gsr_get_stats: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    fetch_counters(c);
    // harvest in submission order so that "last frame" fields end up describing the newest frame
    FrameSlot* last = latest_slot(c);
    for (int k = 0; k < GSR_MAX_SLOTS; ++k)
        if (&c->slot[k] != last) harvest_slot(c, c->slot[k]);
    if (last) {
        harvest_slot(c, *last);
        FrameSlot& sl = *last;
        c->st.n_visible = (int64_t)(sl.h_counters[6] & 0xffffffffull);
        c->st.lazy_redo_tiles = sl.last_lazy ? (int64_t)(sl.h_counters[7] & 0xffffffffull) : 0;
        c->st.pairs_consumed = (int64_t)sl.h_counters[1];
        c->st.entries_scanned = (int64_t)sl.h_counters[3];
        c->st.pairs_total = sl.last_pairs;
        c->st.clusters_total = c->nclus;
        c->st.clusters_kept = sl.surv_hint;
        c->st.policy_bits = (c->lazy_pays ? 1 : 0) | (c->order_pays ? 2 : 0) | (c->cull_pol.pays ? 4 : 0) | (c->cull_pol.weak ? 8 : 0) | (c->prefix_cheaper ? 16 : 0) | (c->depth_active ? 32 : 0);
        c->st.cull_dilate = c->cull_pol.dilate;
        c->st.cull_holdoff = c->cull_pol.holdoff;
        c->st.tiles_x = sl.last_tiles_x;
        c->st.tiles_y = sl.last_local_ty;
        c->st.super_tile = sl.super_tile;
        c->st.stiles_x = sl.stiles_x;
        c->st.stiles_y = sl.stiles_y;
        // running totals live per slot
        int64_t rec_tot = 0, ent_tot = 0, ev_tot = 0;
        for (int k = 0; k < GSR_MAX_SLOTS; ++k) {
            if (!c->slot[k].frame_id) continue;
            rec_tot += (int64_t)c->slot[k].h_counters[2];
            ent_tot += (int64_t)c->slot[k].h_counters[4];
            ev_tot += (int64_t)c->slot[k].h_counters[5];
        }
        c->st.blend_pairs_consumed_total = rec_tot;
        c->st.blend_entries_scanned_total = ent_tot;
        c->st.blend_wave_evals_total = ev_tot;
        {   // colours evaluated ahead of time (device-side running counters, one per slot)
            unsigned long long tot = 0, v = 0;
            for (int k = 0; k < GSR_MAX_SLOTS; ++k)
                if (c->slot[k].lazy_ctr && hipMemcpy(&v, c->slot[k].lazy_ctr + 1, 8, hipMemcpyDeviceToHost) == hipSuccess) tot += v;
            c->st.lazy_colours_total = (int64_t)tot - c->lazy_base;
        }
    }
    *out = c->st;
    return GSR_OK;
}

extern "C" int gsr_stats_reset(gsr_context* c)
{
    if (!c) return set_err(GSR_E_INVALID, "gsr_stats_reset: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    for (int k = 0; k < GSR_MAX_SLOTS; ++k) {
        harvest_slot(c, c->slot[k]);
        HIP_TRY(hipMemset(c->slot[k].counters, 0, 8 * sizeof(unsigned long long)));
        for (int j = 0; j < 8; ++j) c->slot[k].h_counters[j] = 0;
        HIP_TRY(hipMemset(c->slot[k].d_frame, 0, 8 * sizeof(unsigned long long)));
    }
    const int64_t ns = c->st.n_splats;
    {
        unsigned long long tot = 0, v = 0;
        for (int k = 0; k < GSR_MAX_SLOTS; ++k)
            if (c->slot[k].lazy_ctr && hipMemcpy(&v, c->slot[k].lazy_ctr + 1, 8, hipMemcpyDeviceToHost) == hipSuccess) tot += v;
        c->lazy_base = (int64_t)tot;
    }
    const int64_t ups = c->st.uploads;
    double upm[6];
    std::memcpy(upm, c->st.upload_ms, sizeof upm);      // (the last upload's figures describe the resident cloud: they outlive a reset)
    c->st = gsr_stats{};
    c->st.uploads = ups;
    std::memcpy(c->st.upload_ms, upm, sizeof upm);
    c->st.n_splats = ns;
    c->st.record_bytes = (int32_t)sizeof(GsrRecord);
    c->st.pair_bytes = 8;
    return GSR_OK;
}

// ---------------------------------------------------------------------------
// debug / test access: intermediates of the most recent frame
// number of entries of the depth-sorted list of the last frame (= visible splats of that frame)
static uint32_t sorted_count(FrameSlot* sl)
{
    (void)hipMemcpy(sl->h_counters, sl->d_frame, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    return (uint32_t)(sl->h_counters[6] & 0xffffffffull);
}

// storage slot -> upload index, on the host (debug read-backs speak upload indices); empty = identity
static int host_perm(gsr_context* c)
{
    if (!c->perm || c->h_perm.size() == c->n) return GSR_OK;
    c->h_perm.resize(c->n);
    if (c->n) HIP_TRY(hipMemcpy(c->h_perm.data(), c->perm, (size_t)c->n * 4, hipMemcpyDeviceToHost));
    return GSR_OK;
}
static inline uint32_t to_upload_index(const gsr_context* c, uint32_t slot) { return c->perm && slot < c->h_perm.size() ? (uint32_t)c->h_perm[slot] : slot; }

extern "C" int gsr_debug_read_storage_order(gsr_context* c, int32_t* perm, int64_t n)
{
    if (!c || !perm || n < 0 || (uint64_t)n != c->n) return set_err(GSR_E_INVALID, "gsr_debug_read_storage_order: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    if ((rc = host_perm(c))) return rc;
    for (int64_t j = 0; j < n; ++j) perm[j] = (int32_t)to_upload_index(c, (uint32_t)j);
    return GSR_OK;
}

extern "C" int gsr_debug_read_records(gsr_context* c, gsr_debug_record* out, int64_t n)
{
    if (!c || !out || n < 0 || (uint64_t)n > c->n) return set_err(GSR_E_INVALID, "gsr_debug_read_records: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    int src = sync_all(c);
    if (src) return src;
    FrameSlot* sl = latest_slot(c);
    if (!sl) return set_err(GSR_E_INVALID, "gsr_debug_read_records: no frame rendered yet");
    if (sl->last_lazy && sl->last_supers > 0 && c->n > 0) {
        // lazy colour leaves most records pending: evaluate every list completely before reading them back
        hipLaunchKernelGGL(k_colour_prefix, dim3((unsigned)(sl->last_supers * CL_BLOCKS_PER_LIST)), dim3(CL_THREADS), 0, sl->stream,
                           sl->job.f, sl->pvA, sl->sstart, sl->send, c->prefix_all, (int)std::min<size_t>(sl->pair_cap, (size_t)0x7fffffff),
                           c->colrow, sl->rec, sl->colour_evals);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(sl->stream));
    }
    if ((src = host_perm(c))) return src;
    const uint32_t ns = sorted_count(sl);
    const int64_t n_out = n;
    n = (int64_t)c->n;   // (records live at storage slots: read them all, hand back the caller's first n_out upload indices)
    GsrRecord* hr = new (std::nothrow) GsrRecord[n ? n : 1];
    uint32_t* hk = new (std::nothrow) uint32_t[ns ? ns : 1];
    uint2* hv = new (std::nothrow) uint2[ns ? ns : 1];
    int rc = GSR_OK;
    if (!hr || !hk || !hv) rc = set_err(GSR_E_OOM, "gsr_debug_read_records: host allocation failed");
    hipError_t e = hipSuccess;
    if (!rc && n) {
        e = hipMemcpy(hr, sl->rec, (size_t)n * sizeof(GsrRecord), hipMemcpyDeviceToHost);
        // keys and rects live in depth order and only for the visible splats: un-permute through the payload
        if (e == hipSuccess && ns) e = hipMemcpy(hk, sl->keyA, (size_t)ns * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && ns) e = hipMemcpy(hv, sl->valA, (size_t)ns * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = set_err(GSR_E_HIP, "gsr_debug_read_records: %s", hipGetErrorString(e));
    }
    if (!rc) {
        for (int64_t i = 0; i < n_out; ++i) std::memset(&out[i], 0, sizeof(out[i]));
        for (uint32_t r = 0; r < ns; ++r) {
            const uint32_t slot = hv[r].x;
            if ((int64_t)slot >= n) continue;
            const uint32_t i = to_upload_index(c, slot);
            if ((int64_t)i >= n_out) continue;
            gsr_debug_record& o = out[i];
            const GsrRecord& q = hr[slot];
            o.visible = 1;
            o.cx = q.cx; o.cy = q.cy; o.a1x = q.a1x; o.a1y = q.a1y; o.b1x = q.b1x; o.b1y = q.b1y;
            o.hx = q.hx; o.hy = q.hy; o.r = q.r; o.g = q.g; o.b = q.b; o.la = q.la;
            const uint32_t kb = hk[r] + sl->key_min;   // keys are stored relative to the frame's key_min
            std::memcpy(&o.key, &kb, 4);
        }
    }
    delete[] hr; delete[] hk; delete[] hv;
    return rc;
}

extern "C" int gsr_debug_read_depth_order(gsr_context* c, int32_t* perm, int64_t cap, int64_t* n_sorted)
{
    if (!c || !perm || !n_sorted || cap < 0) return set_err(GSR_E_INVALID, "gsr_debug_read_depth_order: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    FrameSlot* sl = latest_slot(c);
    if (!sl) return set_err(GSR_E_INVALID, "gsr_debug_read_depth_order: no frame rendered yet");
    const uint32_t ns = sorted_count(sl);
    *n_sorted = ns;
    const int64_t m = (int64_t)ns < cap ? (int64_t)ns : cap;
    if (m) HIP_TRY(hipMemcpy2D(perm, 4, sl->valA, 8, 4, (size_t)m, hipMemcpyDeviceToHost));
    if ((rc = host_perm(c))) return rc;
    for (int64_t r = 0; r < m; ++r) perm[r] = (int32_t)to_upload_index(c, (uint32_t)perm[r]);
    return GSR_OK;
}

extern "C" int gsr_debug_read_tile_lists(gsr_context* c, int32_t* list_start, int32_t* list_end, int64_t n_lists,
                                         int32_t* pair_splat, int64_t n_pairs)
{
    if (!c || !list_start || !list_end) return set_err(GSR_E_INVALID, "gsr_debug_read_tile_lists: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    FrameSlot* sl = latest_slot(c);
    if (!sl) return set_err(GSR_E_INVALID, "gsr_debug_read_tile_lists: no frame rendered yet");
    if (n_lists != (int64_t)sl->last_supers || n_pairs != (int64_t)sl->last_pairs || (n_pairs > 0 && !pair_splat))
        return set_err(GSR_E_INVALID, "gsr_debug_read_tile_lists: expected %d lists / %u pairs", sl->last_supers, sl->last_pairs);
    if (n_lists) {
        HIP_TRY(hipMemcpy(list_start, sl->sstart, (size_t)n_lists * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(list_end, sl->send, (size_t)n_lists * 4, hipMemcpyDeviceToHost));
    }
    if (n_pairs) HIP_TRY(hipMemcpy2D(pair_splat, 4, sl->pvA, 8, 4, (size_t)n_pairs, hipMemcpyDeviceToHost));
    if ((rc = host_perm(c))) return rc;
    for (int64_t r = 0; r < n_pairs; ++r) pair_splat[r] = (int32_t)to_upload_index(c, (uint32_t)pair_splat[r] & sl->job.f.idx_mask);   // (depth bits off: gsr_device.h)
    return GSR_OK;
}

extern "C" int gsr_debug_read_tile_work(gsr_context* c, uint32_t* work4, int64_t n_tiles)
{
    if (!c || !work4) return set_err(GSR_E_INVALID, "gsr_debug_read_tile_work: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    FrameSlot* sl = latest_slot(c);
    if (!sl || n_tiles != (int64_t)sl->last_tiles_x * sl->last_local_ty)
        return set_err(GSR_E_INVALID, "gsr_debug_read_tile_work: expected %d tiles", sl ? sl->last_tiles_x * sl->last_local_ty : 0);
    if (n_tiles) HIP_TRY(hipMemcpy(work4, sl->tile_work, (size_t)n_tiles * 16, hipMemcpyDeviceToHost));
    return GSR_OK;
}

// what the slot's NEXT frame would be culled against, per tile of the whole image: [0] the (dilated) depth horizons (distance^2, +inf = none),
// [1] the raw horizons before dilation with the tile's status in the sign (k_blend.h: GsrHorizonArgs.stat), [2] the dilated status, [3] the
// covered depths of the last depth-tested frame as K1 saw them (level 0 of pyrc, k_cluster.h)
extern "C" int gsr_debug_read_horizons(gsr_context* c, float* out4, int64_t n_tiles)
{
    if (!c || !out4) return set_err(GSR_E_INVALID, "gsr_debug_read_horizons: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    FrameSlot* sl = latest_slot(c);
    if (!sl || n_tiles <= 0) return set_err(GSR_E_INVALID, "gsr_debug_read_horizons: no frame");
    std::vector<float> zero((size_t)n_tiles, 0.0f);
    HIP_TRY(hipMemcpy(out4, sl->hpyr, (size_t)n_tiles * 4, hipMemcpyDeviceToHost));                     // (level 0 starts at offset 0)
    HIP_TRY(hipMemcpy(out4 + n_tiles, sl->hraw, (size_t)n_tiles * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out4 + 2 * n_tiles, sl->hstat, (size_t)n_tiles * 4, hipMemcpyDeviceToHost));
    if (sl->dpyr && sl->dpyr_cap >= (size_t)n_tiles) HIP_TRY(hipMemcpy(out4 + 3 * n_tiles, sl->dpyr + (size_t)(2 * sl->dpar + 1) * sl->dpyr_cap, (size_t)n_tiles * 4, hipMemcpyDeviceToHost));
    else std::memcpy(out4 + 3 * n_tiles, zero.data(), (size_t)n_tiles * 4);
    return GSR_OK;
}

static int debug_sort_pairs(gsr_context* c, uint32_t* keys, uint32_t* vals, int64_t n64, int key_bits, bool local);
extern "C" int gsr_debug_sort_pairs(gsr_context* c, uint32_t* keys, uint32_t* vals, int64_t n64, int key_bits)
{
    return debug_sort_pairs(c, keys, vals, n64, key_bits, false);
}
static thread_local uint32_t dbg_lo = 0;
static thread_local int dbg_shift = 0;
extern "C" int gsr_debug_sort_pairs_local(gsr_context* c, uint32_t* keys, uint32_t* vals, int64_t n64, int key_bits, uint32_t bucket_lo, int bucket_shift)
{
    if (bucket_shift < 0 || bucket_shift > 31) return set_err(GSR_E_INVALID, "gsr_debug_sort_pairs_local: bad bucket shift");
    dbg_lo = bucket_lo; dbg_shift = bucket_shift;
    return debug_sort_pairs(c, keys, vals, n64, key_bits, true);
}
static int debug_sort_pairs(gsr_context* c, uint32_t* keys, uint32_t* vals, int64_t n64, int key_bits, bool local)
{
    if (!c || n64 < 0 || n64 > 0x7fffffffll || key_bits < 1 || key_bits > 32 || (n64 > 0 && (!keys || !vals)))
        return set_err(GSR_E_INVALID, "gsr_debug_sort_pairs: bad argument");
    if (n64 == 0) return GSR_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    FrameSlot& sl = c->slot[0];
    const uint32_t n = (uint32_t)n64;
    uint32_t *kA = nullptr, *kB = nullptr, *vA = nullptr, *vB = nullptr;
    if ((rc = dev_alloc(&kA, n)) || (rc = dev_alloc(&kB, n)) || (rc = dev_alloc(&vA, n)) || (rc = dev_alloc(&vB, n))) {
        dev_free(kA); dev_free(kB); dev_free(vA); dev_free(vB);
        return rc;
    }
    hipError_t e = hipMemcpyAsync(kA, keys, (size_t)n * 4, hipMemcpyHostToDevice, sl.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(vA, vals, (size_t)n * 4, hipMemcpyHostToDevice, sl.stream);
    if (e == hipSuccess) {
        if (local) {   // one scatter into BK_BUCKETS buckets of width 2^shift from lo, then every bucket on its own (k_radix_local)
            // (uint32 payloads in the slot's bucket regions: the payload region is large enough for either type)
            const uint32_t nblk = div_up(n, RS_TILE);
            hipError_t e2 = hipMemsetAsync(sl.bkt_cnt, 0, (size_t)BK_BUCKETS * BK_STRIDE * sizeof(uint32_t), sl.stream);
            if (e2 == hipSuccess) e2 = hipMemsetAsync(sl.d_counts + 2, 0, 4, sl.stream);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bucket_scatter<uint32_t, false>), dim3(nblk), dim3(RS_THREADS), 0, sl.stream, kA, vA, n, (const uint32_t*)nullptr,
                               dbg_shift, dbg_lo, (const uint32_t*)nullptr, sl.bkt_cnt, sl.bkt_key, reinterpret_cast<uint32_t*>(sl.bkt_val), (uint32_t*)nullptr, sl.d_counts + 2);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_radix_local<uint32_t>), dim3(BK_BUCKETS), dim3(RL_THREADS), 0, sl.stream, sl.bkt_cnt, dbg_shift, key_bits, dbg_lo,
                               sl.bkt_key, reinterpret_cast<uint32_t*>(sl.bkt_val), kA, vA, (uint32_t*)nullptr, (uint32_t*)nullptr);
            uint32_t over = 0;
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(&over, sl.d_counts + 2, 4, hipMemcpyDeviceToHost, sl.stream);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(sl.stream);
            if (e2 != hipSuccess) rc = set_err(GSR_E_HIP, "gsr_debug_sort_pairs_local: %s", hipGetErrorString(e2));
            else if (over) rc = set_err(GSR_E_INVALID, "gsr_debug_sort_pairs_local: a bucket overflowed its region of %d keys (the pipeline would fall back to the global sort)", BK_CAP);
        } else {
            rc = radix_sort(sl, kA, vA, kB, vB, n, key_bits, true, (uint32_t*)nullptr, RS_XCD_DEPTH != 0);
        }
        if (!rc) {
            e = hipMemcpyAsync(keys, kA, (size_t)n * 4, hipMemcpyDeviceToHost, sl.stream);
            if (e == hipSuccess) e = hipMemcpyAsync(vals, vA, (size_t)n * 4, hipMemcpyDeviceToHost, sl.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(sl.stream);
        }
    }
    dev_free(kA); dev_free(kB); dev_free(vA); dev_free(vB);
    if (e != hipSuccess) return set_err(GSR_E_HIP, "gsr_debug_sort_pairs: %s", hipGetErrorString(e));
    return rc;
}

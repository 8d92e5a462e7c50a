// k_blend.h -- K6: per-tile front-to-back compositing with in-kernel tile filtering.
//
// Replaces the reference's fragment shader
// (/root/reference/gsplat_plugin/shaders/GSplatShaderSource.h:304-312) and the
// fixed-function blend src=ONE_MINUS_DST_ALPHA, dst=ONE
// (src/GSplatRenderer.C:613-621): per pixel, nearest first,
//     C += (1-A) * rgb*alpha ;  A += (1-A) * alpha .
//
// Geometry: one 256-thread workgroup per 16x16 tile; wave w owns the 8x8
// quadrant (w&1, w>>1), lane l the pixel (l&7, l>>3) of it -- a wave64 is
// exactly one 8x8 pixel block, so all culling is wave-uniform.
// Data flow (scan -> every wave on its own: gather, test, stage, composite): the tile walks the depth-ordered list of its
// SUPER-tile 1024 entries at a time (splat index + tile mask, 8 B each, coalesced, prefetched one step ahead); entries whose
// mask contains this tile are COMPACTED, in list order, into an LDS hit queue (wave ballots + one count per wave through
// LDS).  A batch of <= BL_BATCH queued hits is then handled by every wave ON ITS OWN, in sub-rounds of 64: lane l gathers
// the l-th hit's 48-B record (3 x dwordx4; the four waves load the same lines), forms its two affine forms at the tile
// origin, tests it against THIS wave's 8x8 quadrant only (bbox + separating axes), and the survivors go, by ballot rank =
// still in depth order, into the wave's own LDS list, two records interleaved per block so that the operands of the
// packed-FP32 inner loop arrive as register pairs.  No barrier between the batch barriers.
// Early-out: a PIXEL stops accumulating once 1-A < 2^-14 (dropped contribution
// <= 2^-14 * max colour, inside the 1e-3 budget; being per pixel it does not
// depend on chunking, so sharded and unsharded frames are bit-identical); a wave
// stops when all its pixels have, and the workgroup stops reading its list when
// all four waves have.
#pragma once
#include <type_traits>
#include "gsr_device.h"
#include "k_preprocess.h"   // gsr_splat_colour_from_row (on-demand colour of the lazy path)
#include "k_cluster.h"      // the depth-horizon pyramid

#ifndef BL_ROUND
#define BL_ROUND 64           // records a wave gathers, tests and stages at a time (= lanes)
#endif
#ifndef BL_WAVES_PER_EU
#define BL_WAVES_PER_EU 6     // 6 waves per SIMD: the kernel needs 68 VGPRs, no scratch
#endif
#ifndef BL_CHECK
#define BL_CHECK 4            // pair iterations between two "is the wave opaque?" tests inside a list (a power of two)
#endif
#ifndef BL_BATCH
#define BL_BATCH 256          // queued hits handed to the waves between two workgroup barriers (a multiple of 64): measured on
                              // C4 64 -> 0.195 ms, 128 -> 0.190, 256 -> 0.187.  Requesting the next sub-round's records ahead of
                              // the inner loop (12 more registers: 5 waves per SIMD) measured 0.196: latency is covered already
#endif
#define BL_QCAP 2048          // hit-queue ring capacity (>= BL_BATCH + the 1024 entries of a scan step)
#define BL_PAIR_F4_MAX 6      // float4s per staged record PAIR: 5 (80 B), 6 with the depth test
#define GSR_T_MIN 6.103515625e-05f  // 2^-14

// BL_PROFILE (variant builds only, tools/blend_phases.py): per-wave shader-clock time of each phase, summed over the launch
#ifdef BL_PROFILE
#define BLP_MAX_WG 65536
__device__ unsigned long long g_blend_prof[BLP_MAX_WG][4][16];   // per workgroup and wave: no atomics (they would dominate the run)
#define BLP(i) { const long long now_ = clock64(); prof[i] += (unsigned long long)(now_ - tlast); tlast = now_; }
#else
#define BLP(i)
#endif

// a tile's work in (roughly) instructions: wave-record evaluations, records gathered, list entries scanned
__device__ __forceinline__ uint32_t gsr_tile_weight(const uint4& w) { return w.z * 32u + w.y * 8u + (w.x >> 1); }

#ifdef GSR_DEBUG_XCC
__device__ uint32_t g_dbg_xcc[4];
#endif
struct GsrBlendArgs {
    int32_t width, height;      // full image
    int32_t tiles_x;            // tiles per row
    int32_t local_tiles;        // tiles_x * local_tiles_y
    GsrShard shard;             // which tile rows this launch owns
    int32_t band_rows;          // pixel rows of the output (band) image
    int32_t super_shift;        // log2(super-tile edge in tiles)
    int32_t rect_shift;         // the list entries' column / row bits are in units of (1 << rect_shift) tiles
    int32_t stiles_x;
    int32_t use_map;            // blockIdx -> tile through tile_map (XCD-aware order)
    int32_t flags;              // GSR_FLAG_*
    int32_t list_cap;           // entries the list buffer holds (a speculative launch may see ranges beyond it)
    uint32_t* sup_work;         // [256] work per super-tile, summed over its tiles (or NULL)
    // front-slab frames (gsr_api.hip): phase 1 (slab = 1) also stores every pixel's transmittance T in tbuf -- the alpha channel
    // holds 1 - T, which does not give T back bit for bit -- and phase 2 (slab = 2) CONTINUES from the stored (C, T): a tile that
    // phase 1 left opaque (bit 0 of its tile_work_a entry) is finished, every other tile composites the splats beyond the slab on
    // top of what it has.  The pixels are those of the one-pass frame, bit for bit: the same records in the same order.
    int32_t slab;
    float* tbuf;                // [band pixels]
    const uint4* tile_work_a;   // phase 2: phase 1's per-tile bookkeeping
    // Depth-tested frames under a depth buffer that was merely CLEARED (the common case): the host, going by the slot's previous
    // depth-tested frame, launches the PLAIN kernel -- no depth load, no sixth operand vector, 66 registers instead of 80 -- with a
    // guard: the word the frame's depth pyramid pass leaves ("some pixel is covered", k_cluster.h).  A launch whose guess was wrong
    // does nothing (no pixel, no bookkeeping), and the host, which reads the same word with the frame's pair count, queues the
    // depth-tested kernel behind it (gsr_api.hip: frame_finish).  The word is loaded first thing and looked at after the first scan.
    const uint32_t* guard;      // NULL: no guard
    uint32_t guard_want;        // the launch is void unless (*guard != 0) == (guard_want != 0)
    // "met the geometry" (bit 14 of the bookkeeping) with a MARGIN: under a perspective projection window depth is alpha - beta / d (d =
    // view depth), so (alpha - zwin_hit) >= near_scale * (alpha - depth) says "the hit lies at least 12 % nearer than the geometry".  A tile
    // counts as clear of the geometry -- classic, k_tile_pass -- only if every record it gathered was; near_scale = 0: no margin
    // (other projections).  Measured without it on C4 with a sphere just under the cloud's surface: two or three tiles per frame went from
    // "saturated, met nothing" to "cannot saturate" between two frames 3 degrees apart, each one a repaired frame.
    float near_alpha, near_scale;
    const float* tile_cov;      // depth-tested frames with a depth pyramid (k_cluster.h): level 0 of pyrc, [tiles_y][tiles_x]; < 0 = the tile has
                                // no covered pixel.  NULL: not known (every tile loads its depths)
    // list entries whose index word carries a coarse window depth (GsrFrame.idx_mask, gsr_zq): a tile drops, while it SCANS, the entries
    // behind the largest depth under its live pixels -- tile_dmax (level 0 of the depth pyramid: the largest depth under the tile) to begin
    // with, then what its waves report at the end of every batch.  idx_mask = 0xffffffff: no such bits in the lists.
    uint32_t idx_mask;
    float zq0, zqs;
    const float* tile_dmax;     // NULL: no scan-time depth filter
    // Host-target frames (gsr_api.hip: queue_blend): the launch is issued once per BAND of tile rows, each followed by the copy of its
    // rows back to the host on a second stream, so that all but the first band's compositing hides behind the link.  A workgroup whose
    // tile row (local) is outside [row_lo, row_hi) leaves at once.
    int32_t row_lo, row_hi;
};

// Staging layout: one list PER QUADRANT (= per wave), holding the round's records that reach that
// quadrant, in depth order, two records interleaved per block so that every ds_read_b128 lands as
// ready-made operand pairs of the packed-FP32 instructions (no v_mov shuffling):
//   f4 0: a.a1x b.a1x a.a1y b.a1y     f4 3: a.r a.g a.b a.la
//   f4 1: a.b1x b.b1x a.b1y b.b1y     f4 4: b.r b.g b.b b.la
//   f4 2: a.c0  b.c0  a.c1  b.c1      f4 5: a.zwin b.zwin - -          (HAS_DEPTH only)
// c0/c1 = the two affine forms of the record at the TILE origin (contract v2): kq0 = lx*a1x + ly*a1y + c0 for the
// pixel (lx, ly) of the tile -- computed once per (record, tile) by the gathering thread.
// HAS_DEPTH = false compiles the depth compare and the sixth float4 out of the inner loop.
#if BL_WAVES_PER_EU > 0
#define BL_OCC __attribute__((amdgpu_waves_per_eu(BL_WAVES_PER_EU)))
#else
#define BL_OCC
#endif
// Lazy colour (k_colour.h): a record whose colour is still pending makes the plain kernel (LAZY = false) give the tile up
// -- it is appended to `redo` -- and the LAZY = true instantiation, launched right behind, composites exactly those tiles
// with the colour evaluated on demand in the gather.
struct GsrLazyArgs {
    GsrFrame f;
    const uint4* colrow;       // 128-byte row per splat: position + colour halves
    int32_t* redo;             // tiles given up by the plain kernel
    uint32_t* redo_count;
};
template <bool HAS_DEPTH, bool LAZY>
__device__ __forceinline__ void
gsr_blend_tile(const GsrBlendArgs& a, const int32_t* __restrict__ tile_map, const uint2* __restrict__ svals,
               const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
               const GsrRecord* __restrict__ recs, float4* __restrict__ out, uint4* __restrict__ tile_work,
               const float* __restrict__ zwin, const float* __restrict__ depth, const GsrLazyArgs& lz)
{
    constexpr int PF4 = 5;
    __shared__ float4 slist[4][(BL_ROUND / 2) * PF4];
    // depth-tested frames: the staged records' window depths, pair by pair, in an array of their OWN -- as a sixth vector of the pair
    // block they cost every tile of the launch (a 96-byte block stride puts the staging stores on 8 banks instead of 16, and the plain
    // loop of a tile the geometry does not touch skipped 16 of every 96 bytes): the depth-tested kernel under a clear buffer was 10 % slower
    __shared__ float2 szl[HAS_DEPTH ? 4 : 1][HAS_DEPTH ? BL_ROUND / 2 : 1];
    __shared__ uint32_t q[BL_QCAP];   // hit queue: splat indices in list (= depth) order
    __shared__ __attribute__((aligned(16))) uint32_t scnt[2][4];   // hits of each wave in a scan step
    __shared__ uint32_t sdone[2][4];
    __shared__ float swmax[2][4];     // depth-tested frames: per wave, the largest depth under its pixels that are not opaque yet (batch end)
    __shared__ uint32_t sevals, sredo, sdmet;
    __shared__ uint32_t slastu[4];    // depth-tested frames: the same count at the moment the wave's UNCOVERED pixels were all opaque
    __shared__ uint32_t stail[32];    // hits queued after each of the last 32 scan steps (ring): which step held a given hit?
    __shared__ uint32_t slast[4];     // per wave: how many of the tile's hits it has gathered (exclusive count)

    // Workgroup b runs on XCD b%8 (observed dispatch order, MI355X guide): tile_map hands each
    // XCD whole super-tiles, whose 64 tiles read the same list and gather the same records.
    int tile;
    if (LAZY) {
        if (blockIdx.x >= *lz.redo_count) return;
        tile = lz.redo[blockIdx.x];
    } else {
        tile = a.use_map ? tile_map[blockIdx.x] : (int)blockIdx.x;
    }
#ifdef GSR_DEBUG_XCC
    if (threadIdx.x == 0) {   // does workgroup b really run on XCD b % 8?
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xfu;
        atomicAdd(&g_dbg_xcc[(xcc == (blockIdx.x & 7u)) ? 0 : 1], 1u);
        if (blockIdx.x == 0) g_dbg_xcc[2] = xcc;
    }
#endif
    if (tile < 0 || tile >= a.local_tiles) return;
    if (a.row_hi != 0x7fffffff) { const int row = tile / a.tiles_x; if (row < a.row_lo || row >= a.row_hi) return; }   // (banded launches only)
    // (a guarded launch: the word is requested here and looked at once, below, where the list's bounds -- requested at the same time -- are
    //  needed anyway.  Looked at INSIDE the loop, as at first, it cost every launch 4 us: a scalar load waits on the counter the LDS shares)
    const uint32_t guard_now = (!HAS_DEPTH && a.guard) ? *a.guard : 0u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef BL_PROFILE
    unsigned long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // lane occupancy of the record evaluations (is an 8x8-pixel wave the wrong granularity for small splats?): lanes whose fragment is
    // inside the quad AND above 1/255 (and passes the depth test), evaluations, and evaluations that touch only ONE 4-row half of the quadrant
    unsigned long long occ_lanes = 0, occ_evals = 0, occ_half = 0, occ_none = 0;
    long long tlast = clock64();
    const unsigned long long wall0 = wall_clock64();   // constant 100 MHz, the same on every CU
#endif
    const int tx = tile % a.tiles_x, lty = tile / a.tiles_x;
    const int gty = gsr_shard_global_row(a.shard, lty);
    const int px = tx * GSR_TILE_PX + (wave & 1) * 8 + (lane & 7);
    const int py = gty * GSR_TILE_PX + (wave >> 1) * 8 + (lane >> 3);
    const bool pix_ok = (px < a.width) && (py < a.height);
    // pixel index inside the tile (contract v2: fragment positions are relative to the tile origin)
    const gsr_v2f lx = (gsr_v2f)((float)((wave & 1) * 8 + (lane & 7))), ly = (gsr_v2f)((float)((wave >> 1) * 8 + (lane >> 3)));
    // first pixel centre of the tile, and the centre of this wave's quadrant box of pixel centres
    const float tcx0 = (float)(tx * GSR_TILE_PX) + 0.5f, tcy0 = (float)(gty * GSR_TILE_PX) + 0.5f;
    const float qcx = tcx0 + 3.5f + 8.0f * (float)(wave & 1), qcy = tcy0 + 3.5f + 8.0f * (float)(wave >> 1);
    if (tid == 0) { sevals = 0; sredo = 0; sdmet = 0; }
    if (tid < 4) slast[tid] = 0u;
    uint32_t my_evals = 0;            // (wave-uniform) records this wave evaluated for its 64 pixels
    uint32_t my_last = 0;             // (wave-uniform) hits of the tile this wave has gathered so far: all of the sub-rounds it entered
    // depth test against what the opaque pass left (depth writes stay off): a fragment survives iff its quad's
    // window depth <= depth[pixel] (src/GSplatRenderer.C:595-610).  No depth buffer = +inf.
    // (a tile without a covered pixel -- the depth pyramid pass knows -- is an ordinary tile: its 256 depths are not even loaded)
    const bool tile_plain = HAS_DEPTH && a.tile_cov != nullptr && !(a.flags & GSR_FLAG_NO_DEPTH_CLASS) && a.tile_cov[gty * a.tiles_x + tx] < 0.0f;   // (uniform)
    const float dpx = (HAS_DEPTH && pix_ok && !tile_plain) ? depth[(size_t)py * a.width + px] : __builtin_inff();
    // Per WAVE (= 8x8 quadrant), once: what the opaque pass left under it.  K1 keeps a splat only if -w <= z <= w, so every window
    // depth in the lists is <= 1 (an IEEE quotient of z <= w is <= 1, and fma(q, 0.5, 0.5) of q <= 1 is <= 1): a quadrant whose
    // depth buffer was CLEARED TO THE FAR PLANE (all >= 1) passes every fragment -- d_triv -- and runs the plain loop, bit-identical
    // by construction.  Otherwise w_max (per sub-round, below) / d_wmin = the largest depth under the quadrant's pixels that are not opaque
    // yet / the smallest depth under the quadrant (a NaN pixel fails every fragment: -inf): a record whose window depth exceeds w_max can
    // pass nowhere it could contribute and is not staged; one at or below d_wmin passes everywhere and needs no per-pixel compare; only
    // the records in between take the depth-tested loop.
    // (formed right before the first gather, not here: `dpx` is a load, and consuming it here would put its latency in front of the
    //  list loads below instead of beside them -- measured: +13 us per C4 launch, five generations of tiles x ~2 us)
    bool d_met = false;               // (wave-uniform) a record that reaches this quadrant did not pass the depth test everywhere in it
    bool d_triv = true, d_init = !HAS_DEPTH;
    float d_wmin = __builtin_inff();

    const int st = (gty >> a.super_shift) * a.stiles_x + (tx >> a.super_shift);
    const int s = sstart[st];
    const int e_ = send[st] < a.list_cap ? send[st] : a.list_cap;
    const int n = e_ > s ? e_ - s : 0;
    if (!HAS_DEPTH && a.guard && (guard_now != 0u) != (a.guard_want != 0u)) return;   // (uniform) a void launch: the other kernel draws this frame

    gsr_v2f C01 = {0.0f, 0.0f};   // {C0, C1} as a register pair
    float C2 = 0.0f, T = 1.0f;    // blue, transmittance 1 - A
    bool wave_done = false;
    if (a.slab == 2) {
        // front-slab phase 2: finished tiles keep what they have; the others go on from the stored colour and transmittance
        const uint4 wa = a.tile_work_a[tile];
        if (wa.w & 1u) {
            if (tid == 0) tile_work[tile] = make_uint4(0u, 0u, 0u, 1u | 0x8000u | (0xffffu << 16));
            return;
        }
        if (pix_ok) {
            const size_t at = (size_t)(lty * GSR_TILE_PX + (wave >> 1) * 8 + (lane >> 3)) * a.width + (size_t)(tx * GSR_TILE_PX + (wave & 1) * 8 + (lane & 7));
            const float4 c = out[at];
            C01 = (gsr_v2f){c.x, c.y}; C2 = c.z; T = a.tbuf[at];
        }
        wave_done = __all(!pix_ok || T < GSR_T_MIN);
    }
    // Depth-tested frames: a pixel the opaque pass COVERED (depth < 1) may never saturate -- whatever lies in front of the geometry is
    // all it can ever get, and K1 keeps exactly that for it (k_preprocess.h: zwin <= the largest covered depth of the tile).  What
    // the depth horizons are about is the tile's UNCOVERED pixels: u_done = they are all opaque, my_last_u = how many of the tile's
    // hits the wave had gathered by then.  (No depth buffer: u_done is wave_done.)
    bool u_done = false;
    uint32_t my_last_u = 0;
    uint32_t es_u = 0xffu;            // (uniform) scan steps that hold what the tile's uncovered pixels needed (0xff: unknown / not reached)
    bool u_all = false;               // (uniform) every wave's uncovered pixels are opaque
    uint32_t fetched = 0;             // (uniform) queued hits handed to the waves


    // ---- scan state (identical in every thread)
    int scan_pos = 0;                 // next list entry to scan
    uint32_t q_head = 0, q_tail = 0;  // monotonic; slot = counter & (BL_QCAP-1)
    int spar = 0;
    // A list entry is (splat index, tile mask): bit c of the low half = the splat's rect reaches column c of the
    // super-tile, bit 16 + r = row r, in rect units (k_bin_place).  This tile is in the rect iff both of ITS bits are set.
    // (uniform) entries whose index word exceeds zlim lie behind every live pixel of the tile: code(zwin) > code(limit) => zwin > limit
    uint32_t zlim = 0xffffffffu;
    const bool zfilter = HAS_DEPTH && a.tile_dmax != nullptr && a.idx_mask != 0xffffffffu && !tile_plain && !(a.flags & GSR_FLAG_NO_DEPTH_CLASS);
    if (zfilter) zlim = (gsr_zq(a.tile_dmax[gty * a.tiles_x + tx], a.zq0, a.zqs) << GSR_ZQ_SHIFT) | a.idx_mask;
    // (the depth bits of the index words: only lists of depth-tested frames carry any, and only the depth-tested kernel reads such lists --
    //  gsr_api.hip: queue_back_end.  Held in a register: re-loaded from the kernel arguments inside the gather it cost 2 us per launch)
    uint32_t imask = HAS_DEPTH ? a.idx_mask : 0xffffffffu;
    if (HAS_DEPTH) asm volatile("" : "+s"(imask));
    const uint32_t sub_mask = (1u << a.super_shift) - 1u;
    const uint32_t tile_bits = (1u << ((tx & sub_mask) >> a.rect_shift)) | (0x10000u << ((gty & sub_mask) >> a.rect_shift));
    // thread t scans entries 4t .. 4t+3 of a 1024-entry step: two 16-byte loads, prefetched one step ahead
    typedef uint32_t gsr_u4 __attribute__((ext_vector_type(4), aligned(8)));
    gsr_u4 preA = {0u, 0u, 0u, 0u}, preB = {0u, 0u, 0u, 0u};
    auto prefetch = [&]() __attribute__((always_inline)) {
        const int i = scan_pos + 4 * tid;
        if (i < n) {   // (may read up to three entries past the list's end: masked below; the buffer is padded)
            const gsr_u4* src = reinterpret_cast<const gsr_u4*>(svals + s + i);
            preA = src[0]; preB = src[1];
        }
    };
    prefetch();
    int round = 0;
    int first_hit_step = -1;          // (uniform) scan step (1024 entries each) that queued the tile's first hit
    bool saturated = false;           // left because every pixel is opaque, not because the list ended

    // How deep did the tile have to LOOK for its first `c` hits?  Not as deep as it has scanned (the scan runs a batch of hits and a
    // prefetched step ahead): they sit in the first scan step after which at least that many hits were queued (the ring remembers the
    // last 32 steps).  All threads call it; returns the number of scan steps (0: none needed, 0xff: unknown / too many).
    auto ext_steps_of = [&](uint32_t c) __attribute__((always_inline)) -> uint32_t {
        const int cur = (scan_pos >> 10) - 1;            // the last step that was scanned
        if (c == 0u) return 0u;
        if (cur < 0) return 0xffu;
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (the lane id afresh: two instructions, and nothing derived from it stays live across the loop)
        const bool reached = ln < 32 && cur - ln >= 0 && stail[(cur - ln) & 31] >= c;
        const unsigned long long m = __ballot(reached);   // bit l: after step cur - l the queue already held the hit
        const int k = __builtin_ctzll(~m | (1ull << 63)); // (bit 0 is always set in a wave that sees the ring: everything was queued by the last step)
        const int e = cur - (k - 1) + 1;
        return (m & 1ull) == 0ull || e <= 0 || e > 0xfe ? 0xffu : (uint32_t)e;
    };
    auto init_depth = [&]() __attribute__((always_inline)) {
        d_init = true;
        u_done = __all(!pix_ok || T < GSR_T_MIN || dpx < 1.0f);
        d_triv = __all(dpx >= 1.0f) && !(a.flags & GSR_FLAG_NO_DEPTH_CLASS);
        if (a.flags & GSR_FLAG_NO_DEPTH_CLASS) {   // (A/B and test hook: every record is staged and compared per pixel, as before round 6)
            d_wmin = -__builtin_inff();
        } else if (!d_triv) {
            const float ninf = -__builtin_inff();
            float vmin = pix_ok ? (dpx == dpx ? dpx : ninf) : __builtin_inff();
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) vmin = __builtin_fminf(vmin, __shfl_xor(vmin, d, 64));
            d_wmin = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, vmin)));
        }
    };
    BLP(0)
    for (;;) {
        // (1) SCAN until a batch of hits is queued, or the list ends
        bool scanned_any = false;
        constexpr int batch = BL_BATCH;
        while ((int)(q_tail - q_head) < batch && scan_pos < n) {
            const int rem = n - (scan_pos + 4 * tid);   // entries of this thread that exist
            bool h0 = rem > 0 && (preA.y & tile_bits) == tile_bits, h1 = rem > 1 && (preA.w & tile_bits) == tile_bits;
            bool h2 = rem > 2 && (preB.y & tile_bits) == tile_bits, h3 = rem > 3 && (preB.w & tile_bits) == tile_bits;
            if (HAS_DEPTH) { h0 = h0 && preA.x <= zlim; h1 = h1 && preA.z <= zlim; h2 = h2 && preB.x <= zlim; h3 = h3 && preB.z <= zlim; }
            const unsigned long long b0 = __ballot(h0), b1 = __ballot(h1), b2 = __ballot(h2), b3 = __ballot(h3);
            // hits of the lower lanes: v_mbcnt accumulates, eight instructions for the four ballots
            uint32_t pos = 0;
            pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, pos));
            pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, pos));
            pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, pos));
            pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b3, pos));
            const uint32_t wtot = (uint32_t)(__builtin_popcountll(b0) + __builtin_popcountll(b1) + __builtin_popcountll(b2) +
                                             __builtin_popcountll(b3));
            if (lane == 0) scnt[spar][wave] = wtot;
            __syncthreads();
            const uint4 wt = *reinterpret_cast<const uint4*>(scnt[spar]);
            pos += q_tail + (wave > 0 ? wt.x : 0u) + (wave > 1 ? wt.y : 0u) + (wave > 2 ? wt.z : 0u);
            if (h0) { q[pos & (BL_QCAP - 1)] = preA.x; ++pos; }
            if (h1) { q[pos & (BL_QCAP - 1)] = preA.z; ++pos; }
            if (h2) { q[pos & (BL_QCAP - 1)] = preB.x; ++pos; }
            if (h3) { q[pos & (BL_QCAP - 1)] = preB.z; }
            q_tail += wt.x + wt.y + wt.z + wt.w;
            if (tid == 0) stail[(scan_pos >> 10) & 31] = q_tail;
            if (first_hit_step < 0 && q_tail != 0u) first_hit_step = scan_pos >> 10;
            scan_pos += 1024;
            spar ^= 1;
            scanned_any = true;
            prefetch();
        }
        if (scanned_any) __syncthreads();            // queue writes visible to every wave
        BLP(1)

        // (2) every wave on its own from here to the end of the batch: the queued hits in sub-rounds of 64.  Lane l takes
        // the l-th hit, loads its record (the four waves load the same lines: L1 hits), tests it against THIS wave's 8x8
        // quadrant only, and the surviving records go -- still in depth order, by ballot rank -- into the wave's own LDS list.
        // No other wave reads that list, so nothing between here and the end of the batch needs a workgroup barrier.
        const int avail = (int)(q_tail - q_head);
        const int take = avail < batch ? avail : batch;
        if (take == 0) break;                         // list exhausted and queue empty
        fetched += (uint32_t)take;
        BLP(2)
        if (HAS_DEPTH && !d_init) init_depth();
        if (!wave_done) {
            for (int sub = 0; sub < take; sub += 64) {
                my_last = q_head + (uint32_t)(sub + 64 < take ? sub + 64 : take);
                const bool have = sub + lane < take;
                // (depth-tested quadrants that hold geometry) what still matters are the pixels that are not opaque yet: a record behind
                // the depth of every one of THEM passes nowhere it could contribute (an opaque pixel takes nothing more).  A quadrant on the geometry's silhouette whose uncovered pixels have gone opaque thus
                // stops staging at the geometry, instead of compositing the rest of its list into pixels that cannot take it.
                float w_max = __builtin_inff(), w_min = -__builtin_inff();   // (GSR_FLAG_NO_DEPTH_CLASS: every record is staged and compared per pixel)
                if (HAS_DEPTH && !d_triv && !(a.flags & GSR_FLAG_NO_DEPTH_CLASS)) {
                    const bool live = pix_ok && T >= GSR_T_MIN;
                    const float dl = dpx == dpx ? dpx : -__builtin_inff();
                    float vmax = live ? dl : -__builtin_inff();
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) vmax = __builtin_fmaxf(vmax, __shfl_xor(vmax, d, 64));
                    w_max = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, vmax)));
                    w_min = d_wmin;     // (the smallest depth under ALL of the quadrant's pixels: the live ones' would do, and costs two registers too many)
                }
                float4 r1, r2;
                float rz = 0.0f, c0 = 0.0f, c1 = 0.0f;
                bool hit = false, pending = false, dtest = false, dmet_l = false;
                if (have) {
                    const uint32_t ridx = q[(q_head + (uint32_t)(sub + lane)) & (BL_QCAP - 1)] & imask;
                    const float4* p = reinterpret_cast<const float4*>(recs + ridx);
                    const float4 r0 = p[0];
                    r1 = p[1]; r2 = p[2];
                    rz = (HAS_DEPTH && !d_triv) ? zwin[ridx] : 0.0f;
                    // Can the splat touch this quadrant?  Separating-axis test of the oriented quad (shrunk to the radius
                    // where alpha can still reach 1/255) against the quadrant's box of pixel centres: the box axes (= bbox
                    // test) and the quad's own two axes.  Conservative.
                    // r0 = (cx, cy, hx, hy), r1 = (a1x, a1y, b1x, b1y) = kappa e/s1, kappa e_perp/s2, r2 = (r, g, b, la = log2 opacity)
                    const float rqk = (((a.flags & GSR_FLAG_NO_ALPHA_RADIUS) ? 2.0f : gsr_support_radius_from_la(r2.w)) + 1.0e-3f) * GSR_KAPPA;
                    const float ddx = qcx - r0.x, ddy = qcy - r0.y;
                    // (hx, hy carry K1's own margin of 1e-4 relative + 0.01 px)
                    const bool box = __builtin_fmaxf(__builtin_fabsf(ddx) - r0.z, __builtin_fabsf(ddy) - r0.w) <= 3.5f;
                    // radius of the quadrant's box of pixel centres along each (scaled) quad axis
                    const float lim1 = rqk + 3.5f * (__builtin_fabsf(r1.x) + __builtin_fabsf(r1.y));
                    const float lim2 = rqk + 3.5f * (__builtin_fabsf(r1.z) + __builtin_fabsf(r1.w));
                    const float pu = __builtin_fabsf(ddx * r1.x + ddy * r1.y);
                    const float pv = __builtin_fabsf(ddx * r1.z + ddy * r1.w);
                    hit = box && (((a.flags & GSR_FLAG_NO_SAT) != 0) || (pu <= lim1 && pv <= lim2));
                    if (HAS_DEPTH && !d_triv) {
                        // (reaches the quadrant, but not CLEARLY in front of everything under it)
                        dmet_l = hit && !(rz <= d_wmin && (a.near_alpha - rz) >= a.near_scale * (a.near_alpha - d_wmin));
                        hit = hit && (rz <= w_max);           // behind everything the opaque pass left under the quadrant's live pixels: no fragment passes
                        dtest = hit && !(rz <= w_min);        // in front of all of it: every fragment passes
                    }
                    pending = __builtin_bit_cast(uint32_t, r2.x) == GSR_COLOUR_PENDING;
                    if (LAZY) {
                        if (hit && pending) gsr_splat_colour_from_row(lz.f, lz.colrow, ridx, r2.x, r2.y, r2.z);
                    }
                    // the record's two affine forms at the tile origin (contract v2, same operations as the oracle)
                    const float d0x = tcx0 - r0.x, d0y = tcy0 - r0.y;
                    c0 = gsr_fma(d0x, r1.x, d0y * r1.y);
                    c1 = gsr_fma(d0x, r1.z, d0y * r1.w);
                }
                if (!LAZY && __any(pending)) {   // the colour pass did not reach this record: the tile goes to the LAZY instantiation
                    if (lane == 0) sredo = 1u;   // (every wave that is still compositing sees the same records)
                    wave_done = true;
                    break;
                }
                const unsigned long long bal = __ballot(hit);
                const int cnt = (int)__builtin_popcountll(bal);     // wave-uniform (scalar)
                const bool dslow = HAS_DEPTH && __any(dtest);       // (uniform) some staged record needs the per-pixel depth compare
                if (HAS_DEPTH && __any(dmet_l)) d_met = true;
                if (hit) {
                    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));   // hits of the lower lanes
                    float* blk = reinterpret_cast<float*>(&slist[wave][(pos >> 1) * PF4]);
                    const uint32_t h = pos & 1u;
                    blk[0 + h] = r1.x; blk[2 + h] = r1.y; blk[4 + h] = r1.z; blk[6 + h] = r1.w;
                    blk[8 + h] = c0; blk[10 + h] = c1;
                    reinterpret_cast<float4*>(blk)[3 + h] = r2;
                    if (HAS_DEPTH) reinterpret_cast<float*>(&szl[wave][pos >> 1])[h] = rz;
                }
                if ((cnt & 1) && lane == 0) {   // odd list: pad with a record that cannot contribute (la = -inf: discarded everywhere)
                    float* blk = reinterpret_cast<float*>(&slist[wave][(cnt >> 1) * PF4]);
                    blk[1] = 0.0f; blk[3] = 0.0f; blk[5] = 0.0f; blk[7] = 0.0f;
                    blk[9] = 0.0f; blk[11] = 0.0f;
                    // colour 0, la -inf (the axes may be stale garbage: a NaN there is rejected by the quad test)
                    {   // (volatile scalar stores: as one float4 constant the compiler kept it live across the whole kernel -- and spilt it)
                        // (... and through an LDS pointer: a volatile store through a GENERIC one is a flat_store + s_waitcnt each -- round 6 shipped
                        //  that for half a day: +5 % on the plain kernel, found in an A/B against round 5's tree on one box)
                        typedef volatile __attribute__((address_space(3))) float lds_vfloat;
                        lds_vfloat* vb = (lds_vfloat*)blk;
                        vb[16] = 0.0f; vb[17] = 0.0f; vb[18] = 0.0f; vb[19] = -__builtin_inff();
                    }
                    if (HAS_DEPTH) reinterpret_cast<float*>(&szl[wave][cnt >> 1])[1] = 0.0f;
                }
                // the list is written and read by this wave only: LDS operations of one wave execute in order
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                BLP(3)

                // (3) composite: the wave walks its list, two records per iteration in packed FP32
                const int npairs = (cnt + 1) >> 1;
                const float4* L = slist[wave];
                struct PairOps { float4 v0, v1, v2, v3, v4; float2 z; };
                auto load_pair = [&](auto with_depth, int p) __attribute__((always_inline)) {
                    PairOps o;
                    o.v0 = L[p * PF4 + 0]; o.v1 = L[p * PF4 + 1]; o.v2 = L[p * PF4 + 2]; o.v3 = L[p * PF4 + 3]; o.v4 = L[p * PF4 + 4];
                    if constexpr (decltype(with_depth)::value) o.z = szl[wave][p]; else o.z = make_float2(0.0f, 0.0f);
                    return o;
                };
                auto blend_ops = [&](auto with_depth, const PairOps& o, gsr_v2f& C01, float& C2, float& T) __attribute__((always_inline)) {
                    const float4 v0 = o.v0, v1 = o.v1, v2 = o.v2, v3 = o.v3, v4 = o.v4;
                    // kappa * (quad-local coordinate) of this pixel for the two records
                    const gsr_v2f q0 = gsr_fma2(lx, (gsr_v2f){v0.x, v0.y}, gsr_fma2(ly, (gsr_v2f){v0.z, v0.w}, (gsr_v2f){v2.x, v2.y}));
                    const gsr_v2f q1 = gsr_fma2(lx, (gsr_v2f){v1.x, v1.y}, gsr_fma2(ly, (gsr_v2f){v1.z, v1.w}, (gsr_v2f){v2.z, v2.w}));
                    const gsr_v2f pw = gsr_fma2(q0, q0, q1 * q1);
                    // contract v3: alpha = clamp(2^(la - pw), 0, 1) on the transcendental unit (v_exp_f32, the clamp is its output
                    // modifier); the fragment is discarded iff la - pw < -log2(255) -- decided on the argument, which the oracle
                    // forms with the same operations (lanes outside the quad may feed 2^x anything: their result is unused)
                    const float arga = v3.w - pw.x, argb = v4.w - pw.y;
                    const float ala = __builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_exp2f(arga), 0.0f), 1.0f);
                    const float alb = __builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_exp2f(argb), 0.0f), 1.0f);
                    bool ina = (__builtin_fmaxf(__builtin_fabsf(q0.x), __builtin_fabsf(q1.x)) <= GSR_QLIM) && (arga >= -GSR_LOG2_255);
                    bool inb = (__builtin_fmaxf(__builtin_fabsf(q0.y), __builtin_fabsf(q1.y)) <= GSR_QLIM) && (argb >= -GSR_LOG2_255);
                    if constexpr (decltype(with_depth)::value) {
                        ina = ina && (o.z.x <= dpx);
                        inb = inb && (o.z.y <= dpx);
                    }
#ifdef BL_PROFILE
                    {
                        const unsigned long long ba = __ballot(ina), bb = __ballot(inb);
                        occ_lanes += (unsigned long long)(__builtin_popcountll(ba) + __builtin_popcountll(bb));
                        occ_evals += 2ull;
                        occ_half += (unsigned long long)(((ba != 0ull) && (((ba & 0xffffffffull) == 0ull) || ((ba >> 32) == 0ull))) ? 1 : 0) +
                                    (unsigned long long)(((bb != 0ull) && (((bb & 0xffffffffull) == 0ull) || ((bb >> 32) == 0ull))) ? 1 : 0);
                        occ_none += (unsigned long long)((ba == 0ull) ? 1 : 0) + (unsigned long long)((bb == 0ull) ? 1 : 0);
                    }
#endif
                    // branch-free under-blend: a rejected fragment blends weight 0, which leaves C and T
                    // bit-identical and costs no exec-mask juggling on the scalar unit.  w = (1-A)*alpha once;
                    // {C0,C1} update as a register pair.
                    // a PIXEL stops accumulating once T < 2^-14: per pixel, so the image is a pure function of the
                    // depth-ordered records -- independent of rounds, quadrant masks, super-tiles and shards
                    const float wa = T * ((ina && T >= GSR_T_MIN) ? ala : 0.0f);
                    C01 = gsr_fma2((gsr_v2f)(wa), (gsr_v2f){v3.x, v3.y}, C01);
                    C2 = gsr_fma(wa, v3.z, C2);
                    T = T - wa;
                    const float wb = T * ((inb && T >= GSR_T_MIN) ? alb : 0.0f);
                    C01 = gsr_fma2((gsr_v2f)(wb), (gsr_v2f){v4.x, v4.y}, C01);
                    C2 = gsr_fma(wb, v4.z, C2);
                    T = T - wb;
                };
                typedef std::integral_constant<bool, false> no_depth_t;
                typedef std::integral_constant<bool, true> depth_t;
                auto blend_pair = [&](int p, gsr_v2f& C01, float& C2, float& T) __attribute__((always_inline)) { blend_ops(no_depth_t(), load_pair(no_depth_t(), p), C01, C2, T); };
                auto blend_pair_d = [&](int p, gsr_v2f& C01, float& C2, float& T) __attribute__((always_inline)) { blend_ops(depth_t(), load_pair(depth_t(), p), C01, C2, T); };
#ifdef BL_EXP_DOUBLE   // experiment: the inner loop a second time on shadow accumulators (its marginal cost = the time difference)
                {
                    gsr_v2f sC01 = C01; float sC2 = C2, sT = T;
                    for (int p2 = 0; p2 < npairs; ++p2) blend_pair(p2, sC01, sC2, sT);
                    if (sT == 123.0f) { C2 += sC2 + sC01.x; }
                }
#endif
                // (the wave looks every BL_CHECK pairs whether its pixels are all opaque: on average it goes opaque half-way
                //  through a list, and the rest of that list -- ~6 of the tile's ~60 pair iterations -- would be wasted)
                // BL_CHECK pairs per trip, straight-line (one loop branch and one "all opaque?" ballot per BL_CHECK pairs: written pair by
                // pair the compiler left four branches and twenty scalar instructions in every iteration), then the remainder
                int p = 0;
                if (!HAS_DEPTH || !dslow) {   // (no staged record needs the depth compare: the plain loop -- a weight-0 blend and a skipped one leave C, T bit-identical)
                    const int nfull = npairs & ~(BL_CHECK - 1);
                    bool stop = false;
                    for (; p < nfull && !stop; p += BL_CHECK) {
#pragma unroll
                        for (int u = 0; u < BL_CHECK; ++u) blend_pair(p + u, C01, C2, T);
                        stop = __all(!pix_ok || T < GSR_T_MIN);
                    }
                    if (!stop)
                        for (; p < npairs; ++p) blend_pair(p, C01, C2, T);
                } else {   // (the depth-tested form carries a sixth operand vector per pair: BL_CHECK_D pairs per trip -- four spill)
#ifndef BL_CHECK_D
#define BL_CHECK_D 1
#endif
                    const int nfull = npairs & ~(BL_CHECK_D - 1);
                    bool stop = false;
                    for (; p < nfull && !stop; p += BL_CHECK_D) {
#pragma unroll
                        for (int u = 0; u < BL_CHECK_D; ++u) blend_pair_d(p + u, C01, C2, T);
                        stop = __all(!pix_ok || T < GSR_T_MIN);
                    }
                    if (!stop)
                        for (; p < npairs; ++p) blend_pair_d(p, C01, C2, T);
                }
                my_evals += (uint32_t)(2 * p < cnt ? 2 * p : cnt);
                BLP(4)
                if (HAS_DEPTH && !u_done && __all(!pix_ok || T < GSR_T_MIN || dpx < 1.0f)) { u_done = true; my_last_u = my_last; }
                if (__all(!pix_ok || T < GSR_T_MIN)) { wave_done = true; break; }
                __builtin_amdgcn_wave_barrier();   // (the next sub-round overwrites the list)
            }
        }
        q_head += (uint32_t)take;
        const int rpar = round & 1;
        const int wv = __builtin_amdgcn_readfirstlane(wave);   // (a scalar: as a vector the LDS addresses below are formed in the prologue and kept -- or spilt)
        if (HAS_DEPTH && wave_done && !u_done) { u_done = true; my_last_u = my_last; }
        if (HAS_DEPTH && zfilter) {   // (uniform) the largest depth under this wave's pixels that can still take something
            const bool live = pix_ok && T >= GSR_T_MIN && !wave_done;
            float vmax = live ? (dpx == dpx ? dpx : -__builtin_inff()) : -__builtin_inff();
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) vmax = __builtin_fmaxf(vmax, __shfl_xor(vmax, d, 64));
            if (lane == 0) swmax[rpar][wv] = vmax;
        }
        if (lane == 0) { sdone[rpar][wv] = (wave_done ? 1u : 0u) | ((HAS_DEPTH ? u_done : wave_done) ? 2u : 0u); slast[wv] = my_last; if (HAS_DEPTH) slastu[wv] = my_last_u; }
        ++round;
        BLP(5)
        __syncthreads();   // the consumed queue slots may be overwritten from here on; every wave's verdict is in
        BLP(6)
        if (!LAZY && sredo) {   // (uniform: written before the barrier)
            int t3 = tile;
            asm volatile("" : "+s"(t3));   // (or its vector copy is made in the prologue, kept for the whole kernel -- and spilt)
            if (tid == 0 && lz.redo) lz.redo[atomicAdd(lz.redo_count, 1u)] = t3;
            return;
        }
        if (HAS_DEPTH && zfilter) {
            const float tm = __builtin_fmaxf(__builtin_fmaxf(swmax[rpar][0], swmax[rpar][1]), __builtin_fmaxf(swmax[rpar][2], swmax[rpar][3]));
            zlim = (gsr_zq(tm, a.zq0, a.zqs) << GSR_ZQ_SHIFT) | a.idx_mask;
        }
        const uint32_t dall = sdone[rpar][0] & sdone[rpar][1] & sdone[rpar][2] & sdone[rpar][3];
        if (HAS_DEPTH && !u_all && (dall & 2u)) {   // (uniform) the moment the tile's uncovered pixels were all opaque: how deep had it looked?
            u_all = true;
            es_u = ext_steps_of(max(max(slastu[0], slastu[1]), max(slastu[2], slastu[3])));
        }
        if ((dall & 1u) != 0u) { saturated = true; break; }
    }
    if (pix_ok) {
        // (pixel coordinates re-derived from an opaque copy of the thread id: keeping them live across the loop costs a spill)
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int w2 = t2 >> 6, l2 = t2 & 63;
        const int brow = lty * GSR_TILE_PX + (w2 >> 1) * 8 + (l2 >> 3);
        const int bcol = tx * GSR_TILE_PX + (w2 & 1) * 8 + (l2 & 7);
        out[(size_t)brow * a.width + bcol] = make_float4(C01.x, C01.y, C2, 1.0f - T);
        if (a.slab == 1) a.tbuf[(size_t)brow * a.width + bcol] = T;
    }
    BLP(8)
#ifdef BL_PROFILE
    if (lane == 0 && blockIdx.x < BLP_MAX_WG) {
        unsigned long long* o = g_blend_prof[blockIdx.x][wave];
#pragma unroll
        for (int i = 0; i < 9; ++i) o[i] = prof[i];
        o[9] = 1ull; o[10] = (unsigned long long)round; o[11] = wall0; o[12] = wall_clock64();
        o[13] = occ_lanes; o[14] = occ_evals; o[15] = occ_half | (occ_none << 32);
    }
#endif
    // bookkeeping for the roofline: list entries scanned and records gathered by this tile
    __syncthreads();  // orders the sevals = 0 store when the list was empty
    if (lane == 0) { atomicAdd(&sevals, my_evals); if (HAS_DEPTH && d_met) sdmet = 1u; }
    __syncthreads();
    if (HAS_DEPTH && !d_init) init_depth();      // (a tile without a single hit)
    const uint32_t es_c = ext_steps_of(max(max(slast[0], slast[1]), max(slast[2], slast[3])));   // (everything the tile gathered)
    if (!HAS_DEPTH) { u_all = saturated; es_u = es_c; }
    else if (!u_all) {   // (the list ended first: the last batch's verdicts)
        const uint32_t dl = sdone[(round - 1) & 1][0] & sdone[(round - 1) & 1][1] & sdone[(round - 1) & 1][2] & sdone[(round - 1) & 1][3];
        if (round > 0 ? (dl & 2u) != 0u : __syncthreads_and(u_done ? 1 : 0) != 0) { u_all = true; es_u = round > 0 ? ext_steps_of(max(max(slastu[0], slastu[1]), max(slastu[2], slastu[3]))) : 0u; }
    }
    if (tid == 0) {
        // entries actually read: everything up to scan_pos plus the prefetched step
        const int rd = scan_pos + 1024;
        // (.w: bit 0 = every pixel went opaque; bits 1..13 = the 1024-entry scan step that held the tile's first hit; bit 15 = the tile's
        //  UNCOVERED pixels all went opaque (no depth buffer: bit 0 again); bits 16..23 = the number of scan steps that hold everything
        //  the tile gathered, bits 24..31 = ... everything it had gathered when bit 15 came true; 0xff = unknown / too many: fall back to .x)
        // (bit 14: depth-tested frames -- the tile MET the opaque geometry: a record that reached one of its quadrants did not pass the
        //  depth test everywhere in it.  Such a tile saturates, if at all, AT the geometry: k_tile_pass does not count on it next frame)
        const uint32_t fs = (uint32_t)(first_hit_step < 0 ? 0 : (first_hit_step > 0x1fff ? 0x1fff : first_hit_step)) | (sdmet ? 0x2000u : 0u);
        const uint4 tw = make_uint4((uint32_t)(rd < n ? rd : n), fetched, sevals, (saturated ? 1u : 0u) | (fs << 1) | (u_all ? 0x8000u : 0u) | (es_c << 16) | (es_u << 24));
        tile_work[tile] = tw;   // (k_sum_work turns these into colour prefixes, depth horizons and the frame's culling verdict)
        // work of the tile's super-tile, for k_tile_order (fire and forget: ~60 tiles per address and frame)
        if (a.sup_work && gsr_tile_weight(tw)) atomicAdd(&a.sup_work[st], gsr_tile_weight(tw));
    }
}

// the two entry points: the plain kernel at 6 waves per SIMD; the fallback carries the SH evaluation and is left to the
// register allocator (it runs for the few tiles the colour pass did not cover)
template <bool HAS_DEPTH>
__global__ void __launch_bounds__(256) BL_OCC
k_blend(GsrBlendArgs a, const int32_t* __restrict__ tile_map, const uint2* __restrict__ svals,
        const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
        const GsrRecord* __restrict__ recs, float4* __restrict__ out, uint4* __restrict__ tile_work,
        const float* __restrict__ zwin, const float* __restrict__ depth, GsrLazyArgs lz)
{
    gsr_blend_tile<HAS_DEPTH, false>(a, tile_map, svals, sstart, send, recs, out, tile_work, zwin, depth, lz);
}
template <bool HAS_DEPTH>
__global__ void __launch_bounds__(256)
k_blend_lazy(GsrBlendArgs a, const int32_t* __restrict__ tile_map, const uint2* __restrict__ svals,
             const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
             const GsrRecord* __restrict__ recs, float4* __restrict__ out, uint4* __restrict__ tile_work,
             const float* __restrict__ zwin, const float* __restrict__ depth, GsrLazyArgs lz)
{
    gsr_blend_tile<HAS_DEPTH, true>(a, tile_map, svals, sstart, send, recs, out, tile_work, zwin, depth, lz);
}

// The end of a frame, two small kernels.
//   k_tile_pass       one WAVEFRONT per 8x8 block of tiles, lane = tile.  From the blend kernel's per-tile bookkeeping it forms
//                     the DEPTH HORIZON of every tile for the slot's next frame, checks THIS frame's culling tile by tile, and
//                     leaves per-block partial sums of the counters.
//   k_sum_work        one workgroup: partial sums -> the frame's counters; the culling verdict to the host; the lazy-colour
//                     prefixes and the hints (lazy colour / tile order / occlusion culling pay?) for the next frames.
//   k_horizon_dilate  one wavefront per 8x8 block of tiles (Morton order inside the block): every tile's horizon becomes the
//                     largest of its (2r+1)^2 neighbourhood -- the view moves between frames -- and the levels of the max
//                     pyramid of k_cluster.h are wave shuffles.  Queued BEHIND k_sum_work: it runs while the host is still
//                     reacting to the verdict.
// (One workgroup doing the tile walk as well took 36 us at 1080p -- a single CU issuing a few thousand instructions for each
//  of 16 waves; spread over the chip the walk is three dependent round trips.)
// counters[1] (records gathered, this frame), [2] (records, running total), [3] (entries scanned, this frame), [4] (entries,
// running total), [5] (wave-record evaluations, running total): a handful of plain stores instead of atomics per tile
// (same-address atomics serialise at ~12 ns each on MI355X).
#ifdef SW_PROFILE
__device__ unsigned long long g_sw_prof[8];
#define SWP(i) { if (threadIdx.x == 0) g_sw_prof[i] = clock64(); }
#else
#define SWP(i)
#endif
#define SW_THREADS 256
#ifndef SW_HEADROOM_SHIFT
#define SW_HEADROOM_SHIFT 2
#endif
#ifndef SW_HEADROOM_ADD
#define SW_HEADROOM_ADD 1024u
#endif
struct GsrSumArgs {
    int32_t n_tiles, tiles_x, tiles_y /* whole image */, super_shift, stiles_x, n_super;
    GsrShard shard;
};
// Depth horizons (occlusion culling).  A tile that went opaque after scanning `rd` entries of its super-tile's list will, in
// the slot's next frame, need nothing behind the entry a quarter (+1024 entries) further down: that entry's distance^2 is the
// tile's horizon (+inf for a tile that stayed open).  K1 and k_cluster_cull drop what lies beyond the horizon of every tile
// it can reach, so every list still holds, for each of its tiles, all splats in front of that tile's horizon -- and a tile of
// a culled frame is complete iff it went opaque without scanning past the horizon its splats were compared with (at least
// that of the tile's own dilated neighbourhood).  That is checked per tile, from the key of the last entry the tile scanned;
// a frame that fails is rendered again without culling (gsr_api.hip).  Two pyramids alternate: the frame's last kernels
// read the one the frame was culled against while they write the next frame's.
struct GsrHorizonArgs {
    float* raw;                     // out: per-tile horizons for the slot's next frame (k_horizon_dilate makes the pyramid); NULL = off
    const float* pyr_in;            // the pyramid this frame was culled against (if `culled`)
    float* pyr_out;                 // the pyramid k_horizon_dilate fills for the next frame (the same buffer: k_tile_pass is done with it)
    int32_t pyr_off[GSR_PYR_LEVELS];
    int32_t culled;                 // K1 / k_cluster_cull dropped splats behind `pyr_in`
    int32_t dilate;                 // ... after widening every rect by this many tiles
    const uint2* lists;             // the super-tile lists
    int32_t list_cap;               // entries the list buffer holds
    const float4* geoA;             // xyz = position
    float cam[3];
    unsigned long long* host_end;   // mapped host word: ticket << 32 | "a horizon broke"
    uint32_t ticket;
    int32_t slab;                   // 2 = the end of a front-slab frame: tiles that phase 1 left opaque were dealt with by k_slab_mid
    const uint4* tile_work_a;       // ... phase 1's per-tile bookkeeping (added to the frame's counters)
    // Depth-tested frames.  A tile's status, per frame: CLASSIC = every pixel went opaque and none of them met the opaque geometry
    // -- next frame it is treated like any tile (its horizon speaks for all of its pixels); otherwise its horizon speaks for its
    // UNCOVERED pixels only and K1 keeps, on top, what lies in front of the geometry under its covered ones (k_preprocess.h).
    // The raw horizons carry the status in their sign bit (set = not classic); the dilation leaves, per tile, "every tile of the
    // neighbourhood was classic" in `stat`: the depth pyramid pass of the next frame masks the covered depths with it, and that
    // frame's k_tile_pass reads it back to know which of the two promises each tile has to keep.
    float* stat;                    // [tiles_y][tiles_x] 1 / 0; NULL: no depth-tested frames so far (every tile classic)
    int32_t stat_in_use;            // this frame's K1 applied `stat` (a culled depth-tested frame)
    uint32_t idx_mask;              // the list entries' index bits (GsrFrame.idx_mask)
    int32_t depth_culled;           // this frame's K1 dropped splats behind the opaque geometry where no horizon applied (k_preprocess.h)
    uint32_t* dbg;                  // GSR_DEBUG_VIOL in the environment: [0] = tiles that broke their promise, then 8 words for each of the first eight
};
// per-block partial sums of k_tile_pass
struct __attribute__((aligned(16))) GsrTilePartial {
    unsigned long long scanned, fetched, evals;
    uint32_t wmax, unsat, nused, nfin, viol, pad_;
};
__global__ void __launch_bounds__(64)
k_tile_pass(const uint4* __restrict__ tile_work, GsrSumArgs g, const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
            GsrHorizonArgs hz, GsrTilePartial* __restrict__ partial /* [blocks] */,
            uint32_t* __restrict__ st_scan /* [512] per super-tile: deepest scan of its opaque tiles / a tile stayed open (cleared by k_sum_work) */)
{
    const int lane = threadIdx.x;
    const int lx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), ly = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int nbx = (g.tiles_x + 7) >> 3;
    const int b = (int)blockIdx.x, by = b / nbx, bx = b - by * nbx;
    const int tx = bx * 8 + lx, gty = by * 8 + ly;
    const bool inside = tx < g.tiles_x && gty < g.tiles_y;
    if (hz.raw && b == 0) {   // levels 4 and 5 of the NEXT frame's pyramid (the other buffer) are max-reduced by the dilation's workgroups: clear them
        const int n45 = gsr_pyr_dim(g.tiles_x, 4) * gsr_pyr_dim(g.tiles_y, 4) + gsr_pyr_dim(g.tiles_x, 5) * gsr_pyr_dim(g.tiles_y, 5);
        for (int i = lane; i < n45; i += 64) hz.pyr_out[hz.pyr_off[4] + i] = 0.0f;
    }
    bool own = false;
    int ti = 0, st = 0;
    if (inside && gsr_shard_owns(g.shard, gty)) {
        const int lty = g.shard.rpb > 0 ? gty - g.shard.index * g.shard.rpb : gty / g.shard.count;
        const int i = lty * g.tiles_x + tx;
        if (i < g.n_tiles) { own = true; ti = i; st = (gty >> g.super_shift) * g.stiles_x + (tx >> g.super_shift); }
    }
    // every phase issues its loads unconditionally (tiles that do not exist read a safe address and discard the value)
    uint4 w = tile_work[ti];
    // front-slab frames: what phase 1 spent on the tile; a tile it left opaque is finished (its next horizon is k_slab_mid's)
    uint4 wa = make_uint4(0u, 0u, 0u, 0u);
    if (hz.slab == 2 && own) wa = hz.tile_work_a[ti];
    const bool a_done = (wa.w & 1u) != 0u;
    if (a_done) w = make_uint4(0u, 0u, 0u, 0u);
    const int s0 = sstart[st];
    int e0 = send[st];
    float hold = __builtin_inff();
    if (hz.raw && hz.culled && inside)   // (at least) the value of the tile itself, widened like every rect: gsr_pyr_max is monotone
        hold = gsr_pyr_max(hz.pyr_in, hz.pyr_off, g.tiles_x, max(tx - hz.dilate, 0), max(gty - hz.dilate, 0),
                           min(tx + hz.dilate, g.tiles_x - 1), min(gty + hz.dilate, g.tiles_y - 1));
    if (!own) { w = make_uint4(0u, 0u, 0u, 0u); hold = __builtin_inff(); }
    e0 = e0 < hz.list_cap ? e0 : hz.list_cap;
    const uint32_t len = (own && e0 > s0) ? (uint32_t)(e0 - s0) : 0u;
    // how deep the tile looked: the scan steps that hold everything it gathered (k_blend), never more than it read
    // Two notions of "opaque" (k_blend): EVERY pixel (bit 0: what the lazy-colour prefixes, the front slab and the counters go by) and
    // every UNCOVERED pixel (bit 15: what the horizons go by -- a pixel the opaque pass covered gets what lies in front of the geometry
    // through K1's depth clause, exactly, whether or not it saturates; without a depth buffer the two are the same)
    const uint32_t es_all = (w.w >> 16) & 0xffu;
    const uint32_t rd_all = (es_all == 0xffu || (es_all << 10) > w.x) ? w.x : (es_all << 10);
    const bool opaque_all = own && (w.w & 1u) != 0u;
    const bool classic_now = opaque_all && (w.w & 0x4000u) == 0u;       // every pixel opaque, and none of them met the geometry on the way
    // which promise did the tile make for THIS frame?  classic: all of its pixels go opaque inside its horizon (no depth clause was
    // applied on its account); otherwise only the uncovered ones have to (the covered ones were given what lies in front of the geometry)
    const bool pred_classic = !(hz.stat && hz.stat_in_use) || !inside || hz.stat[gty * g.tiles_x + tx] != 0.0f;
    const uint32_t es_u = w.w >> 24;
    const uint32_t rd_u = (es_u == 0xffu || (es_u << 10) > w.x) ? w.x : (es_u << 10);
    // The promise that is CHECKED goes by what the tile was predicted to be; the horizon that is LEFT by what it is now (classic: all of
    // its pixels -- next frame no depth clause will be applied on its account, so the horizon has to cover all of them; otherwise its
    // uncovered pixels).  `opaque` = the tile keeps / can make the promise in question.
    const uint32_t rdc = pred_classic ? rd_all : rd_u;                                   // check
    // (a tile predicted classic keeps its promise when ALL of its pixels go opaque -- whether it met the geometry on the way only decides
    //  what it is predicted to be next time)
    const bool opaque_c = own && (pred_classic ? opaque_all : (classic_now || (w.w & 0x8000u) != 0u));
    const uint32_t rd = classic_now ? rd_all : rd_u;                                     // leave
    const bool opaque = own && (classic_now || (w.w & 0x8000u) != 0u);
    const uint32_t first = ((w.w >> 1) & 0x1fffu) << 10;   // list position (1024-entry granularity) of the tile's first hit
    // where the tile's next horizon sits: a quarter of what it scanned from its first hit on (+1024 entries) beyond the scan -- a
    // tile high up in a super-tile over oblique ground starts deep in the shared list, and that part is not its depth range
    const uint32_t want = rd + ((rd > first ? rd - first : 0u) >> 2) + 1024u;
    {   // per super-tile: deepest scan of its opaque tiles, and whether one stayed open.  The tiles of a super-tile are CONSECUTIVE
        // lanes (Morton order): reduce over them first -- atomics that share a cache line serialise like atomics on one address
        // (8160 of them on five lines took 20 us).
        const int gs = g.super_shift < 3 ? g.super_shift : 3;     // the part of a super-tile inside this 8x8 block: 4^gs lanes
        uint32_t m = opaque_all ? rd_all : 0u, open = (own && !opaque_all && !a_done) ? 1u : 0u, any = own ? 1u : 0u;
        for (int d = 1; d < (1 << (2 * gs)); d <<= 1) {
            const uint32_t om = __shfl_xor(m, d, 64), oo = __shfl_xor(open, d, 64), oa = __shfl_xor(any, d, 64);
            m = om > m ? om : m; open |= oo; any |= oa;
        }
        // (the group's first lane may lie outside the image or belong to another rank: its super-tile index comes from its coordinates)
        const int gst = (gty >> g.super_shift) * g.stiles_x + (tx >> g.super_shift);
        if ((lane & ((1 << (2 * gs)) - 1)) == 0 && any && gst < 256 && (tx >> g.super_shift) < g.stiles_x) {
            if (m) atomicMax(&st_scan[gst], m);
            if (open) st_scan[256 + gst] = 1u;
        }
    }
    uint32_t viol = 0u, nused = 0u, nfin = 0u;
    if (hz.raw) {
        const bool ok = opaque && rd > 0u && rd <= len;
        const bool c1 = opaque_c && rdc > 0u && rdc <= len && hold < 3.0e38f, c2 = ok && want < len;
        const uint32_t i1 = hz.lists[c1 ? (uint32_t)s0 + rdc - 1u : 0u].x & hz.idx_mask;  // the last entry the tile scanned (for the promise it made)
        // Depth-tested frames WITHOUT horizons (the first frames, a repair): K1 dropped what lies behind the opaque geometry, so a tile that
        // saturates just in front of it does so on a list that ENDS there -- "too short" for the rule above, and with no old horizon to push
        // out it would get none.  A tile without a horizon costs far more than itself: the pyramid look-up of a cluster's widened rect
        // spans up to 32 x 32 tiles, and one +inf among them keeps the cluster (measured: 273 such tiles, 42 k surviving clusters instead
        // of 14 k).  Its list's last entry, pushed out by 5 %, stands in; the next frame -- culled, its lists no longer cut inside the
        // horizons -- checks itself as always.
        const bool c3 = ok && !c2 && !(hold < 3.0e38f) && hz.depth_culled != 0;
        const uint32_t i2 = hz.lists[c2 ? (uint32_t)s0 + want : (c3 ? (uint32_t)s0 + len - 1u : 0u)].x & hz.idx_mask;      // the entry its next horizon sits at
        const float4 P1 = hz.geoA[c1 ? i1 : 0u], P2 = hz.geoA[c2 ? i2 : 0u];
        // distance^2 = the sort key of k_preprocess.h, same operations
        float klast, hnew;
        { const float dx = P1.x - hz.cam[0], dy = P1.y - hz.cam[1], dz = P1.z - hz.cam[2]; klast = gsr_fma(dz, dz, gsr_fma(dy, dy, dx * dx)); }
        { const float dx = P2.x - hz.cam[0], dy = P2.y - hz.cam[1], dz = P2.z - hz.cam[2]; hnew = gsr_fma(dz, dz, gsr_fma(dy, dy, dx * dx)); }
        float h = 0.0f;                    // outside the image / another rank's tile: nothing is needed there
        if (own) {
            // this frame: a tile with a horizon must have gone opaque without looking past it (a tile whose uncovered pixels needed
            // nothing at all -- the opaque pass covers it -- has looked at nothing)
            const bool none_c = opaque_c && !pred_classic && rdc == 0u;      // (the uncovered pixels were promised, and needed nothing)
            const bool none = opaque && !classic_now && rd == 0u;
            if (hold < 3.0e38f && !none_c && (!opaque_c || !c1 || !(klast <= hold))) viol = 1u;
            if (viol && hz.dbg) {
                const uint32_t k = atomicAdd(hz.dbg, 1u);
                if (k < 8u) {
                    uint32_t* o = hz.dbg + 1 + 8 * k;
                    o[0] = (uint32_t)tx; o[1] = (uint32_t)gty; o[2] = w.w; o[3] = w.x; o[4] = __float_as_uint(hold); o[5] = __float_as_uint(klast);
                    o[6] = (pred_classic ? 1u : 0u) | (opaque_c ? 2u : 0u) | (c1 ? 4u : 0u) | (classic_now ? 8u : 0u); o[7] = (rdc << 8) | (len > 0xffffffu ? 0xffu : 0u); 
                    o[7] = rdc; 
                }
            }
            // next frame: the key a quarter (+1024 entries) beyond the scan; a list too short for that was itself thinned by
            // culling -- then the old horizon is pushed out by 5 % (in distance^2) instead
            h = __builtin_inff();
            if (none) h = 0.0f;
            else if (opaque) {
                if (c2) h = hnew;
                else if (hold < 3.0e38f) h = hold * 1.05f;
                else if (c3) h = hnew * 1.05f;
            }
        }
        // (the sign bit carries the tile's status for the slot's next frame: set = not classic; tiles of other ranks / outside: classic)
        if (inside && !a_done) hz.raw[gty * g.tiles_x + tx] = (own && !classic_now) ? -h : h;
    }
    // would occlusion culling have something to work with?  Judged on every frame, with or without horizons being prepared: a tile
    // that draws anything counts, and counts as "finished early" when it went opaque in the first 70 % of its list
    if (own && len > 0u) { nused = 1u; if (opaque && rd > 0u && rd <= len && (unsigned long long)want * 10ull <= (unsigned long long)len * 7ull) nfin = 1u; }
    if (a_done) { nused = 1u; nfin = 1u; }   // (went opaque inside the front slab)
    unsigned long long sc = (unsigned long long)w.x + wa.x, fe = (unsigned long long)w.y + wa.y, ev = (unsigned long long)w.z + wa.z;
    uint32_t wmax = gsr_tile_weight(w) + gsr_tile_weight(wa), unsat = (own && !a_done && !(w.w & 1u) && (w.y || wa.y)) ? 1u : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        sc += __shfl_down(sc, d, 64); fe += __shfl_down(fe, d, 64); ev += __shfl_down(ev, d, 64); unsat += __shfl_down(unsat, d, 64);
        nused += __shfl_down(nused, d, 64); nfin += __shfl_down(nfin, d, 64);
        { const uint32_t o = __shfl_down(wmax, d, 64); wmax = o > wmax ? o : wmax; }
    }
    const bool any_viol = __ballot(viol != 0u) != 0ull;
    if (lane == 0) {
        GsrTilePartial p;
        p.scanned = sc; p.fetched = fe; p.evals = ev; p.wmax = wmax; p.unsat = unsat; p.nused = nused; p.nfin = nfin;
        p.viol = any_viol ? 1u : 0u; p.pad_ = 0u;
        partial[b] = p;
    }
}

// The middle of a front-slab frame (gsr_api.hip), one wavefront per 8x8 block of tiles like k_tile_pass.  Phase 1 has composited
// the splats up to the slab key; a tile whose pixels are all opaque now needs NOTHING of what lies beyond it -- no prediction, no
// check: level 0 of the pyramid phase 2 culls against = 0 for such a tile (and for tiles of other ranks), +inf for a tile that is
// still open; the coarser levels are maxima over it.  For the tiles that are finished this is also the moment to form the horizon of the
// slot's NEXT frame (k_tile_pass's rule, on phase 1's lists while they exist; a list too short for it was cut by the slab, and
// the slab's own distance, pushed out by 5 %, stands in).
__global__ void __launch_bounds__(64)
k_slab_mid(const uint4* __restrict__ tile_work_a, GsrSumArgs g, const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
           GsrHorizonArgs hz /* raw (may be NULL), lists, list_cap, geoA, cam, pyr_off */,
           float* __restrict__ pyr2 /* all levels written here */, const uint32_t* __restrict__ slab, uint32_t key_min)
{
    const int lane = threadIdx.x;
    const int lx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), ly = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int nbx = (g.tiles_x + 7) >> 3;
    const int b = (int)blockIdx.x, by = b / nbx, bx = b - by * nbx;
    const int tx = bx * 8 + lx, gty = by * 8 + ly;
    const bool inside = tx < g.tiles_x && gty < g.tiles_y;
    bool own = false;
    int ti = 0, st = 0;
    if (inside && gsr_shard_owns(g.shard, gty)) {
        const int lty = g.shard.rpb > 0 ? gty - g.shard.index * g.shard.rpb : gty / g.shard.count;
        const int i = lty * g.tiles_x + tx;
        if (i < g.n_tiles) { own = true; ti = i; st = (gty >> g.super_shift) * g.stiles_x + (tx >> g.super_shift); }
    }
    const uint4 w = tile_work_a[ti];
    const int s0 = sstart[st];
    int e0 = send[st];
    e0 = e0 < hz.list_cap ? e0 : hz.list_cap;
    const bool opaque = own && (w.w & 1u) != 0u;
    {   // the pyramid phase 2 culls against, all levels (k_horizon_dilate's scheme at radius 0: lane = tile in Morton order; levels
        // 4 and 5 were cleared by phase 1's first k_cluster_cull pass)
        float v = inside ? ((own && !opaque) ? __builtin_inff() : 0.0f) : 0.0f;
        if (inside) pyr2[hz.pyr_off[0] + gty * g.tiles_x + tx] = v;
        v = __builtin_fmaxf(v, __shfl_xor(v, 1, 64)); v = __builtin_fmaxf(v, __shfl_xor(v, 2, 64));
        if ((lane & 3) == 0 && inside) pyr2[hz.pyr_off[1] + (gty >> 1) * gsr_pyr_dim(g.tiles_x, 1) + (tx >> 1)] = v;
        v = __builtin_fmaxf(v, __shfl_xor(v, 4, 64)); v = __builtin_fmaxf(v, __shfl_xor(v, 8, 64));
        if ((lane & 15) == 0 && inside) pyr2[hz.pyr_off[2] + (gty >> 2) * gsr_pyr_dim(g.tiles_x, 2) + (tx >> 2)] = v;
        v = __builtin_fmaxf(v, __shfl_xor(v, 16, 64)); v = __builtin_fmaxf(v, __shfl_xor(v, 32, 64));
        if (lane == 0) {
            pyr2[hz.pyr_off[3] + by * nbx + bx] = v;
            atomicMax(reinterpret_cast<uint32_t*>(pyr2) + hz.pyr_off[4] + (by >> 1) * gsr_pyr_dim(g.tiles_x, 4) + (bx >> 1), __float_as_uint(v));
            atomicMax(reinterpret_cast<uint32_t*>(pyr2) + hz.pyr_off[5] + (by >> 2) * gsr_pyr_dim(g.tiles_x, 5) + (bx >> 2), __float_as_uint(v));
        }
    }
    if (hz.raw) {
        const uint32_t len = (own && e0 > s0) ? (uint32_t)(e0 - s0) : 0u;
        const uint32_t es = (w.w >> 16) & 0xffu;      // (a finished tile: every pixel opaque, covered or not)
        const uint32_t rd = (es == 0xffu || (es << 10) > w.x) ? w.x : (es << 10);
        const uint32_t first = ((w.w >> 1) & 0x1fffu) << 10;
        const uint32_t want = rd + ((rd > first ? rd - first : 0u) >> 2) + 1024u;
        const bool c2 = opaque && rd > 0u && rd <= len && want < len;
        const uint32_t i2 = hz.lists[c2 ? (uint32_t)s0 + want : 0u].x & hz.idx_mask;
        const float4 P2 = hz.geoA[c2 ? i2 : 0u];
        const float dx = P2.x - hz.cam[0], dy = P2.y - hz.cam[1], dz = P2.z - hz.cam[2];
        const float hnew = gsr_fma(dz, dz, gsr_fma(dy, dy, dx * dx));
        const uint32_t ka = slab[0];
        const float slab_d2 = ka == 0xffffffffu ? __builtin_inff() : __builtin_bit_cast(float, ka + key_min);
        float h = 0.0f;
        if (own) h = !opaque ? __builtin_inff() : (c2 ? hnew : slab_d2 * 1.05f);
        // (status for the slot's next frame in the sign bit, as in k_tile_pass: a finished tile that met the geometry is not classic)
        if (inside) hz.raw[gty * g.tiles_x + tx] = (own && !(opaque && (w.w & 0x4000u) == 0u)) ? -h : h;
    }
}

__device__ __forceinline__ void
gsr_sum_work(const GsrTilePartial* __restrict__ partial, int nblocks, GsrSumArgs g, unsigned long long* __restrict__ counters,
           const uint32_t* __restrict__ n_visible, unsigned long long* __restrict__ summary /* device [8]: fetched by gsr_get_stats */,
           uint32_t* __restrict__ prefix /* [256] lazy colour: list entries to colour per super-tile, next frame (or NULL) */,
           const uint32_t* __restrict__ redo_count /* tiles the plain blend kernel gave up this frame (or NULL) */,
           uint32_t* __restrict__ colour_evals /* [256] per-list counts of the colour pass, cleared here (or NULL) */,
           unsigned long long* __restrict__ colour_total /* running total of the above */,
           const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
           uint32_t* __restrict__ lazy_hint /* would lazy colour pay for a frame like this one? (read by the next frames) */,
           uint32_t* __restrict__ sup_work_next /* [256] the NEXT frame's per-super-tile work sums: cleared here (or NULL) */,
           GsrHorizonArgs hz, uint32_t* __restrict__ st_scan /* [512] from k_tile_pass: read, cleared */)
{
    SWP(0)
    __shared__ unsigned long long s_sum[3];
    __shared__ uint32_t s_unsat, s_est, s_cev, s_wmax, s_nfin, s_nused, s_viol;
    // everything this workgroup needs from memory is requested up front: every dependent round trip (~2 us) is frame latency
    unsigned long long old2 = 0, old4 = 0, old5 = 0, old_ct = 0;
    uint32_t nvis = 0, nredo = 0, my_cev = 0, my_m = 0;
    int my_s0 = 0, my_e0 = 0;
    if (threadIdx.x == 0) {
        old2 = counters[2]; old4 = counters[4]; old5 = counters[5];
        nvis = *n_visible;
        nredo = redo_count ? *redo_count : 0u;
        if (colour_evals) old_ct = *colour_total;
        s_unsat = 0; s_est = 0; s_cev = 0; s_wmax = 0; s_nfin = 0; s_nused = 0; s_viol = 0;
    }
    if (threadIdx.x < 3) s_sum[threadIdx.x] = 0;
    if (colour_evals) my_cev = colour_evals[threadIdx.x];
    if ((int)threadIdx.x < g.n_super) { my_m = st_scan[threadIdx.x]; my_s0 = sstart[threadIdx.x]; my_e0 = send[threadIdx.x]; }
    unsigned long long sc = 0, fe = 0, ev = 0;
    uint32_t wmax = 0, unsat = 0, nused = 0, nfin = 0, viol = 0;
    for (int i = (int)threadIdx.x; i < nblocks; i += SW_THREADS) {
        const GsrTilePartial p = partial[i];
        sc += p.scanned; fe += p.fetched; ev += p.evals; unsat += p.unsat; nused += p.nused; nfin += p.nfin; viol |= p.viol;
        wmax = p.wmax > wmax ? p.wmax : wmax;
    }
    __syncthreads();
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        sc += __shfl_down(sc, d, 64); fe += __shfl_down(fe, d, 64); ev += __shfl_down(ev, d, 64); unsat += __shfl_down(unsat, d, 64);
        nused += __shfl_down(nused, d, 64); nfin += __shfl_down(nfin, d, 64);
        { const uint32_t o = __shfl_down(wmax, d, 64); wmax = o > wmax ? o : wmax; }
    }
    const bool any_viol = __ballot(viol != 0u) != 0ull;
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_sum[0], sc); atomicAdd(&s_sum[1], fe); atomicAdd(&s_sum[2], ev); atomicAdd(&s_unsat, unsat); atomicMax(&s_wmax, wmax);
        atomicAdd(&s_nused, nused); atomicAdd(&s_nfin, nfin);
        if (any_viol) s_viol = 1u;
    }
    __syncthreads();
    SWP(2)
    // first thing: the frame's verdict to the host, which is waiting for it before it hands the frame over and queues the next one
    if (hz.host_end && threadIdx.x == 0) {
        uint32_t v = s_viol;
        *hz.host_end = ((unsigned long long)hz.ticket << 32) | (unsigned long long)(v ? 1u : 0u);
        __threadfence_system();   // out to the host NOW (without it the word left with the kernel's end: a 13 us bubble before the next frame)
    }
    if (sup_work_next) sup_work_next[threadIdx.x] = 0u;
    {
        const int st = (int)threadIdx.x;
        if (st < g.n_super) {
            st_scan[st] = 0u; st_scan[256 + st] = 0u;   // (for the slot's next frame)
            if (prefix) {
                const uint32_t len = my_e0 > my_s0 ? (uint32_t)(my_e0 - my_s0) : 0u;
                const uint32_t want = my_m + (my_m >> SW_HEADROOM_SHIFT) + SW_HEADROOM_ADD;   // headroom for the next frame's camera move
                atomicAdd(&s_est, want < len ? want : len);   // colour evaluations the lazy pass would make for a frame like this one
                prefix[st] = want;
            }
        }
        if (colour_evals) {
            colour_evals[threadIdx.x] = 0u;
            if (my_cev) atomicAdd(&s_cev, my_cev);
        }
    }
    __syncthreads();
    SWP(4)
    if (threadIdx.x == 0) {
        // Lazy colour pays when the colour pass would evaluate well under half of what eager evaluation does (it gathers
        // rows at random, eager streams them) and (almost) no tile would need the on-demand fallback.  Small or sparse
        // clouds (BASELINE C2, C3) fail one of the two; the 6 M-splat scenes pass both.
        // Bit 1: would a heaviest-first tile order pay (k_tile_order)?  When the heaviest tile alone is more than half of a
        // workgroup slot's fair share of the frame (1536 slots: 6 workgroups on 256 CUs) raster order leaves a long, thin tail
        // (C3: heaviest tile = 0.85 of the share); when the tiles are all alike, no order helps (C4: 0.34) and the kernel is skipped.
        const unsigned long long wsum = s_sum[2] * 32ull + s_sum[1] * 8ull + (s_sum[0] >> 1);
        // ... and the frame must be long enough for the gain to beat the ordering kernel's ~10 us (C2: heaviest = 1.6 of the
        // share, but 37 us of work in all: 6 us gained)
        const uint32_t order_pays = ((unsigned long long)s_wmax * 3072ull > wsum && wsum > 60000000ull) ? 2u : 0u;
        if (lazy_hint) *lazy_hint = ((prefix && (unsigned long long)s_est * 10ull < (unsigned long long)nvis * 4ull &&
                                      s_unsat * 64u <= (uint32_t)g.n_tiles) ? 1u : 0u) | order_pays |
                                    // bit 2: occlusion culling has something to work with: at least 30 % of the tiles that draw anything
                                    // went opaque in the first 70 % of their list (judged on unculled frames; a culled frame keeps the
                                    // verdict it was given)
                                    // bit 3: in a frame like this one the list-prefix colour pass would evaluate fewer colours than one per
                                    // kept splat (a culled frame that still keeps a lot: oblique ground, silhouettes)
                                    ((prefix && (unsigned long long)s_est * 3ull < (unsigned long long)nvis * 2ull) ? 8u : 0u) |
                                    ((hz.culled || ((unsigned long long)s_nfin * 10ull >= (unsigned long long)s_nused * 3ull && s_nused > 0u)) ? 4u : 0u);
        // running totals: plain read-modify-write (a slot's frames are serialised on its stream; nothing else touches them)
        const unsigned long long t2 = old2 + s_sum[1], t4 = old4 + s_sum[0], t5 = old5 + s_sum[2];
        counters[1] = s_sum[1]; counters[2] = t2; counters[3] = s_sum[0]; counters[4] = t4; counters[5] = t5;
        if (colour_evals) *colour_total = old_ct + s_cev;
        // the frame's summary stays in device memory (gsr_get_stats copies 64 bytes after its stream sync): writing it
        // to mapped host memory every frame cost ~10 us of PCIe round trips at the end of the frame
        summary[0] = 0ull; summary[1] = s_sum[1]; summary[2] = t2; summary[3] = s_sum[0]; summary[4] = t4; summary[5] = t5;
        summary[6] = (unsigned long long)nvis;
        summary[7] = (unsigned long long)nredo;
    }
    SWP(5)
}
__global__ void __launch_bounds__(SW_THREADS)
k_sum_work(const GsrTilePartial* __restrict__ partial, int nblocks, GsrSumArgs g, unsigned long long* __restrict__ counters,
           const uint32_t* __restrict__ n_visible, unsigned long long* __restrict__ summary, uint32_t* __restrict__ prefix,
           const uint32_t* __restrict__ redo_count, uint32_t* __restrict__ colour_evals, unsigned long long* __restrict__ colour_total,
           const int32_t* __restrict__ sstart, const int32_t* __restrict__ send, uint32_t* __restrict__ lazy_hint,
           uint32_t* __restrict__ sup_work_next, GsrHorizonArgs hz, uint32_t* __restrict__ st_scan)
{
    gsr_sum_work(partial, nblocks, g, counters, n_visible, summary, prefix, redo_count, colour_evals, colour_total, sstart, send, lazy_hint,
                 sup_work_next, hz, st_scan);
}

// raw = per-tile horizons (k_tile_pass); pyr = the pyramid the slot's next frame culls against: level 0 = every tile's horizon
// widened to the largest of its (2r+1)^2 neighbourhood, levels 1..3 from wave shuffles (lane = tile in Morton order).
__device__ __forceinline__ void
gsr_horizon_dilate(const float* __restrict__ raw, int tiles_x, int tiles_y, int r, const GsrHorizonArgs& hz, float* __restrict__ pyr, const int b,
                   const int lane /* of the ONE wavefront that handles block b */)
{
    const int lx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), ly = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int nbx = (tiles_x + 7) >> 3;
    const int by = b / nbx, bx = b - by * nbx;
    const int tx = bx * 8 + lx, ty = by * 8 + ly;
    const bool inside = tx < tiles_x && ty < tiles_y;
    float v = 0.0f;
    if (inside) {
        const int x0 = max(tx - r, 0), x1 = min(tx + r, tiles_x - 1), y0 = max(ty - r, 0), y1 = min(ty + r, tiles_y - 1);
        bool allc = true;                 // every tile of the neighbourhood was classic (sign bit clear: k_tile_pass)
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                const float rv = raw[y * tiles_x + x];
                v = __builtin_fmaxf(v, __builtin_fabsf(rv));
                allc = allc && (__float_as_uint(rv) >> 31) == 0u;
            }
        pyr[hz.pyr_off[0] + ty * tiles_x + tx] = v;
        if (hz.stat) hz.stat[ty * tiles_x + tx] = allc ? 1.0f : 0.0f;
    }
    v = __builtin_fmaxf(v, __shfl_xor(v, 1, 64)); v = __builtin_fmaxf(v, __shfl_xor(v, 2, 64));
    if ((lane & 3) == 0 && inside) pyr[hz.pyr_off[1] + (ty >> 1) * gsr_pyr_dim(tiles_x, 1) + (tx >> 1)] = v;
    v = __builtin_fmaxf(v, __shfl_xor(v, 4, 64)); v = __builtin_fmaxf(v, __shfl_xor(v, 8, 64));
    if ((lane & 15) == 0 && inside) pyr[hz.pyr_off[2] + (ty >> 2) * gsr_pyr_dim(tiles_x, 2) + (tx >> 2)] = v;
    v = __builtin_fmaxf(v, __shfl_xor(v, 16, 64)); v = __builtin_fmaxf(v, __shfl_xor(v, 32, 64));
    if (lane == 0) {
        pyr[hz.pyr_off[3] + by * nbx + bx] = v;
        // levels 4 and 5 (cleared by k_sum_work, which runs before this kernel): horizons are >= 0, so their bit patterns order like
        // unsigned integers
        atomicMax(reinterpret_cast<uint32_t*>(pyr) + hz.pyr_off[4] + (by >> 1) * gsr_pyr_dim(tiles_x, 4) + (bx >> 1), __float_as_uint(v));
        atomicMax(reinterpret_cast<uint32_t*>(pyr) + hz.pyr_off[5] + (by >> 2) * gsr_pyr_dim(tiles_x, 5) + (bx >> 2), __float_as_uint(v));
    }
}
__global__ void __launch_bounds__(64)
k_horizon_dilate(const float* __restrict__ raw, int tiles_x, int tiles_y, int r, GsrHorizonArgs hz, float* __restrict__ pyr)
{
    gsr_horizon_dilate(raw, tiles_x, tiles_y, r, hz, pyr, (int)blockIdx.x, (int)threadIdx.x);
}

// The end of a frame that leaves horizons, in ONE launch: workgroups 0 .. nblocks - 1 (their first wavefront) dilate the per-tile
// horizons into the slot's next pyramid, workgroup nblocks sums the frame's counters and posts the culling verdict.  Both read what
// k_tile_pass left and nothing of each other (the pyramid is double-buffered: k_tile_pass cleared the top levels of the one written
// here), so the dilation runs beside the sums instead of behind them.
__global__ void __launch_bounds__(SW_THREADS)
k_frame_end(const GsrTilePartial* __restrict__ partial, int nblocks, GsrSumArgs g, unsigned long long* __restrict__ counters,
            const uint32_t* __restrict__ n_visible, unsigned long long* __restrict__ summary, uint32_t* __restrict__ prefix,
            const uint32_t* __restrict__ redo_count, uint32_t* __restrict__ colour_evals, unsigned long long* __restrict__ colour_total,
            const int32_t* __restrict__ sstart, const int32_t* __restrict__ send, uint32_t* __restrict__ lazy_hint,
            uint32_t* __restrict__ sup_work_next, GsrHorizonArgs hz, uint32_t* __restrict__ st_scan, int dilate_r)
{
    if ((int)blockIdx.x == nblocks) {
        gsr_sum_work(partial, nblocks, g, counters, n_visible, summary, prefix, redo_count, colour_evals, colour_total, sstart, send, lazy_hint,
                     sup_work_next, hz, st_scan);
        return;
    }
    if (threadIdx.x >= 64) return;
    gsr_horizon_dilate(hz.raw, g.tiles_x, g.tiles_y, dilate_r, hz, hz.pyr_out, (int)blockIdx.x, (int)threadIdx.x);
}

// Heaviest tiles first.  The blend kernel's workgroups are dispatched in blockIdx order as slots free up; when a frame's tiles
// differ widely in work (BASELINE C3: median workgroup 5 us, heaviest 80 us) raster order leaves the second half of the launch
// nearly empty.  This kernel (end of the frame, launched only when k_sum_work's verdict says it pays) turns the frame's per-tile
// bookkeeping into the NEXT frame's blockIdx -> tile table: super-tiles keep a home XCD (their tiles share a list and records,
// so an L2) -- dealt by descending work in snake order, which also balances the XCDs -- and inside an XCD the tiles run in
// descending order of work.  One workgroup PER XCD builds that XCD's column of the table from its own super-tiles (a single
// workgroup doing all of it took 25-30 us: one CU's issue rate); the per-super-tile work sums come ready-made from the blend
// kernel's workgroups (sup_work, one atomic per tile).
// order[8 * p + x] = the p-th tile of XCD x, -1 = idle; cap = 2 * ceil(n_super / 16) * tiles per super-tile bounds p (a snake
// period of 16 gives every XCD two super-tiles).
#define TO_THREADS 1024
#define TO_LEVELS 128
__device__ __forceinline__ void
gsr_tile_order(const uint4* __restrict__ tile_work, const GsrSumArgs& g, int tiles_y, const uint32_t* __restrict__ sup_work, int cap,
               int32_t* __restrict__ order, const int xcd)
{
    __shared__ uint32_t s_sup[256];
    __shared__ uint32_t s_own[64];             // this XCD's super-tiles
    __shared__ uint32_t s_hist[TO_LEVELS];
    __shared__ uint32_t s_nown, s_wmax;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 256) s_sup[tid] = tid < g.n_super ? sup_work[tid] : 0u;
    if (tid < TO_LEVELS) s_hist[tid] = 0u;
    if (tid == 0) { s_nown = 0u; s_wmax = 0u; }
    __syncthreads();
    if (tid < 256) {   // rank of super-tile `tid` by descending work -> its XCD, snake order; keep the ones that are ours
        bool mine_here = false;
        if (tid < g.n_super) {
            const uint32_t mine = s_sup[tid];
            uint32_t r = 0;
            for (int u = 0; u < g.n_super; ++u) {
                const uint32_t o = s_sup[u];
                r += (o > mine || (o == mine && u < tid)) ? 1u : 0u;
            }
            const uint32_t m = r & 15u;
            mine_here = (int)(m < 8u ? m : 15u - m) == xcd;
        }
        if (mine_here) s_own[atomicAdd(&s_nown, 1u)] = (uint32_t)tid;   // (at most 2 * 16 of 256; their order does not matter)
    }
    __syncthreads();
    const int shift2 = 2 * g.super_shift, emask = (1 << g.super_shift) - 1;
    const int items = (int)s_nown << shift2;
    // item -> local tile index (or -1: outside the image / another rank's row) and its weight
    auto tile_of = [&](int it, uint32_t& w) -> int {
        const int st = (int)s_own[it >> shift2], j = it & ((1 << shift2) - 1);
        const int sy = st / g.stiles_x, sx = st - sy * g.stiles_x;
        const int tx = (sx << g.super_shift) + (j & emask), gty = (sy << g.super_shift) + (j >> g.super_shift);
        w = 0u;
        if (tx >= g.tiles_x || gty >= tiles_y || !gsr_shard_owns(g.shard, gty)) return -1;
        const int lty = g.shard.rpb > 0 ? gty - g.shard.index * g.shard.rpb : gty / g.shard.count;
        const int i = lty * g.tiles_x + tx;
        if (i >= g.n_tiles) return -1;
        w = gsr_tile_weight(tile_work[i]);
        return i;
    };
    // (a thread's first TO_CACHE items stay in registers for the three passes: each pass used to fetch the tiles' bookkeeping again --
    //  three dependent trips to memory in a kernel that is nothing but its one workgroup's latency: 11.4 us of C3's 197 us frame)
    constexpr int TO_CACHE = 5;
    int ci[TO_CACHE];
    uint32_t cw[TO_CACHE];
#pragma unroll
    for (int k = 0; k < TO_CACHE; ++k) { const int it = tid + k * TO_THREADS; ci[k] = -1; cw[k] = 0u; if (it < items) ci[k] = tile_of(it, cw[k]); }
    {
        uint32_t wmax = 0;
#pragma unroll
        for (int k = 0; k < TO_CACHE; ++k) wmax = cw[k] > wmax ? cw[k] : wmax;
        for (int it = tid + TO_CACHE * TO_THREADS; it < items; it += TO_THREADS) { uint32_t w; (void)tile_of(it, w); wmax = w > wmax ? w : wmax; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(wmax, d, 64); wmax = o > wmax ? o : wmax; }
        if (lane == 0 && wmax) atomicMax(&s_wmax, wmax);
    }
    __syncthreads();
    const float scale = s_wmax ? (float)(TO_LEVELS - 2) / (float)s_wmax : 0.0f;
    // heavy = low level = early; level 127 = the tiles without work
    auto level_of = [&](uint32_t w) { const int l = w ? (TO_LEVELS - 2) - (int)((float)w * scale) : TO_LEVELS - 1; return l < 0 ? 0 : l; };
#pragma unroll
    for (int k = 0; k < TO_CACHE; ++k) if (ci[k] >= 0) atomicAdd(&s_hist[level_of(cw[k])], 1u);
    for (int it = tid + TO_CACHE * TO_THREADS; it < items; it += TO_THREADS) {
        uint32_t w;
        if (tile_of(it, w) >= 0) atomicAdd(&s_hist[level_of(w)], 1u);
    }
    __syncthreads();
    if (wave == 0) {   // exclusive scan of the 128 levels (two per lane)
        const uint32_t a0 = s_hist[2 * lane], a1 = s_hist[2 * lane + 1];
        uint32_t x = a0 + a1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        const uint32_t ex = x - (a0 + a1);
        s_hist[2 * lane] = ex; s_hist[2 * lane + 1] = ex + a0;
        if (lane == 63) s_nown = x;   // tiles in this XCD's column
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TO_CACHE; ++k)
        if (ci[k] >= 0) {
            const uint32_t pos = atomicAdd(&s_hist[level_of(cw[k])], 1u);
            if ((int)pos < cap) order[8 * (int)pos + xcd] = ci[k];
        }
    for (int it = tid + TO_CACHE * TO_THREADS; it < items; it += TO_THREADS) {
        uint32_t w;
        const int i = tile_of(it, w);
        if (i >= 0) {
            const uint32_t pos = atomicAdd(&s_hist[level_of(w)], 1u);
            if ((int)pos < cap) order[8 * (int)pos + xcd] = i;
        }
    }
    for (int p = (int)s_nown + tid; p < cap; p += TO_THREADS) order[8 * p + xcd] = -1;
}
__global__ void __launch_bounds__(TO_THREADS)
k_tile_order(const uint4* __restrict__ tile_work, GsrSumArgs g, int tiles_y, const uint32_t* __restrict__ sup_work, int cap,
             int32_t* __restrict__ order)
{
    gsr_tile_order(tile_work, g, tiles_y, sup_work, cap, order, (int)blockIdx.x);
}

// The end of a frame whose tile order is due, in ONE launch (round 5): the eight workgroups of k_tile_order BESIDE the frame's sums
// (and the horizon dilation, where the frame leaves horizons) instead of behind them -- they read the blend kernel's bookkeeping
// and this frame's half of the work sums, the sums clear the OTHER half: nothing of each other.  Workgroups of 1024 threads (the
// tile order's); the others use their first 256 / 64.  C3: k_frame_end 6.8 + k_tile_order 11.4 us in every frame -> one launch.
struct GsrOrderArgs { const uint4* tile_work; int32_t tiles_y; const uint32_t* sup_work; int32_t cap; int32_t* order; };
__global__ void __launch_bounds__(TO_THREADS)
k_frame_end_order(const GsrTilePartial* __restrict__ partial, int nblocks, int ndilate /* nblocks, or 0: the frame leaves no horizons */, GsrSumArgs g,
                  unsigned long long* __restrict__ counters, const uint32_t* __restrict__ n_visible, unsigned long long* __restrict__ summary,
                  uint32_t* __restrict__ prefix, const uint32_t* __restrict__ redo_count, uint32_t* __restrict__ colour_evals,
                  unsigned long long* __restrict__ colour_total, const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
                  uint32_t* __restrict__ lazy_hint, uint32_t* __restrict__ sup_work_next, GsrHorizonArgs hz, uint32_t* __restrict__ st_scan, int dilate_r,
                  GsrOrderArgs oa)
{
    const int b = (int)blockIdx.x;
    if (b < 8) {                       // (first: they are the longest)
        gsr_tile_order(oa.tile_work, g, oa.tiles_y, oa.sup_work, oa.cap, oa.order, b);
        return;
    }
    if (b == 8) {
        if (threadIdx.x >= SW_THREADS) return;
        gsr_sum_work(partial, nblocks, g, counters, n_visible, summary, prefix, redo_count, colour_evals, colour_total, sstart, send, lazy_hint,
                     sup_work_next, hz, st_scan);
        return;
    }
    // (the dilation: one WAVEFRONT per 8x8 block of tiles, sixteen of them per 1024-thread workgroup -- one block per workgroup held 16 x the
    //  wave slots for a sixteenth of the work.  Whole wavefronts leave early here and in the sums' workgroup above: the barriers of
    //  gsr_sum_work count the waves that are still there, which is what the hardware's s_barrier does)
    static_assert(TO_THREADS % 64 == 0 && SW_THREADS % 64 == 0, "early exits are by whole wavefronts");
    const int blk = (b - 9) * (TO_THREADS / 64) + (int)(threadIdx.x >> 6);
    if (blk >= ndilate) return;
    gsr_horizon_dilate(hz.raw, g.tiles_x, g.tiles_y, dilate_r, hz, hz.pyr_out, blk, (int)(threadIdx.x & 63u));
}

// k_binning.h -- coarse binning: depth-ordered splats -> per-SUPER-tile front-to-back lists.
//
// The GL rasteriser did all binning implicitly for the reference (one instanced
// quad per splat, /root/reference/gsplat_plugin/src/GSplatRenderer.C:647).  Here
// only a COARSE (super-tile = SxS tiles, <= 256 of them) list is materialised, and the
// blend kernel filters each list down to its own 16x16 tile on the fly (wave ballots), so it
// stops reading the moment the tile is opaque.  Materialising per-tile lists instead cost
// 75 M pairs x 2 radix passes on the 6 M-splat scene, 93 % of which were never consumed
// (profiles/r1_baseline_v1).
//
// The lists are built by a counting sort that never materialises (key, value) pairs:
//   k_bin_count   per block of BN_TILE depth-ranked splats: how many of its (splat, super-tile)
//                 pairs fall into each super-tile            -> hist[super][block]
//   k_scan_rows   (k_sort.h) exclusive scan of every super-tile's row over the blocks
//   (k_bin_ranges exclusive scan of the per-super-tile totals -> list ranges, pair count D: folded into k_bin_place, a launch of
//                 its own only for a frame that has no list buffer yet)
//   k_bin_place   every block re-derives its pairs and writes (splat index, tile rect) straight
//                 to its final list position, in depth order
// versus emit pairs -> radix pass (histogram + scatter) -> find ranges: 8 B written per pair
// instead of 12 B written + 16 B read + 12 B written, and 4 launches instead of 11.
// Roofline: HBM / L2 write combining (8 B per pair, scattered over <= 256 open lists).
#pragma once
#include "gsr_device.h"
#include "k_sort.h"

#define BN_THREADS 256
#ifndef BN_ITEMS
#define BN_ITEMS 4                         // splats per thread (measured on C4: 2 -> 0.056 ms place + 0.058 count/scan, 4 -> 0.065 + 0.044, 8 -> 0.081 + 0.041)
#endif
#define BN_TILE (BN_THREADS * BN_ITEMS)    // splats per block
#define BN_WAVE_ITEMS (BN_TILE / 4)        // wave w owns the w-th contiguous quarter of the block
#define BN_BINS 256                        // super-tiles are at most 256 (GsrFrame.super is chosen that way)

#define BN_BIG 8                           // a splat covering more super-tiles than this is expanded by the whole wave

// (`shift` in this file = log2 of the super-tile edge in RECT UNITS = GsrFrame.super_shift - rect_shift; sh.g = rect_shift.)
// What a list entry carries instead of the rect: which columns (bits 0..15) and rows (bits 16..31), in rect units, of super-tile
// (sx, sy) the rect reaches.  A tile of the super-tile is inside the rect iff its column bit and its row bit are both set:
// two instructions in the blend kernel's list scan instead of four byte compares.
__device__ __forceinline__ uint32_t bn_tile_mask(uint32_t rc, int sx, int sy, int shift)
{
    const int x0 = rc & 255, y0 = (rc >> 8) & 255, x1 = (rc >> 16) & 255, y1 = rc >> 24;
    const int e = (1 << shift) - 1;
    const int c0 = max(x0 - (sx << shift), 0), c1 = min(x1 - (sx << shift), e);
    const int r0 = max(y0 - (sy << shift), 0), r1 = min(y1 - (sy << shift), e);
    const uint32_t cols = ((2u << c1) - 1u) & ~((1u << c0) - 1u), rows = ((2u << r1) - 1u) & ~((1u << r0) - 1u);
    return cols | (rows << 16);
}

// visit the owned super-tiles of a packed tile rect
// (occlusion culling needs nothing here: K1 keeps a splat iff one of the tiles of its rect may still need it, and then it
//  enters every list its rect reaches -- a tile that does not need it has gone opaque before it gets there)
// (row_lo, row_hi: the super-tile rows this work item owns -- BnPart below; 0 .. INT_MAX where a block is not split)
template <typename F>
__device__ __forceinline__ void bn_for_each_super(uint32_t rc, int shift, const GsrShard& sh, int stiles_x, int row_lo, int row_hi, F&& fn)
{
    const int x0 = rc & 255, y0 = (rc >> 8) & 255, x1 = (rc >> 16) & 255, y1 = rc >> 24;
    if (x1 < x0 || y1 < y0) return;
    const int sx0 = x0 >> shift, sx1 = x1 >> shift;
    for (int sy = max(y0 >> shift, row_lo); sy <= min(y1 >> shift, row_hi); ++sy) {
        if (sh.count > 1) {
            const int lo = max(y0, sy << shift), hi = min(y1, ((sy + 1) << shift) - 1);
            if (gsr_owned_rect_rows(lo, hi, sh) == 0) continue;
        }
        const uint32_t rowkey = (uint32_t)sy * (uint32_t)stiles_x;
        for (int sx = sx0; sx <= sx1; ++sx) fn(rowkey + (uint32_t)sx, sx, sy);
    }
}

// All (splat, super-tile) pairs of a GROUP of 64 depth-consecutive splats (lane = splat, v = its
// (index, rect)): fn(owner_lane, owner_v, super_tile).  The nearest splats cover dozens of
// super-tiles while the median covers one or two, and depth order puts all the big ones into the
// same few groups -- so a lane walks its own rect only when it is small; the rect of a big splat
// is spread over the 64 lanes (the caller's fn never assumes owner_lane == its own lane).
// The NEAREST splats of a view from inside the cloud fill the screen: the first block of the depth order then holds a thousand
// splats of 135 super-tiles each, and one workgroup expanded them all (R1: workgroup 0 of k_bin_place 126 us, the median one 15).
// The first BN_SPLIT_TILES blocks are therefore each handled by one workgroup PER ROW of super-tiles: a part walks only the cells
// of its row (hist columns and list positions of different super-tiles never meet, so the parts need not know of each other).
#ifndef BN_SPLIT_TILES
#define BN_SPLIT_TILES 16
#endif
struct BnPart { uint32_t tile; int row_lo, row_hi; };
// work item v of nv = nb + min(nb, BN_SPLIT_TILES) * (rows - 1) -> (block, rows); blocks behind the split ones go to the XCDs in contiguous eighths
__device__ __forceinline__ uint32_t bn_items(uint32_t nb, int rows) { const uint32_t hs = nb < (uint32_t)BN_SPLIT_TILES ? nb : (uint32_t)BN_SPLIT_TILES; return nb + hs * (uint32_t)(rows - 1); }
__device__ __forceinline__ BnPart bn_part(uint32_t v, uint32_t nb, int rows)
{
    const uint32_t hs = nb < (uint32_t)BN_SPLIT_TILES ? nb : (uint32_t)BN_SPLIT_TILES;
    if (v < hs * (uint32_t)rows) { const uint32_t t = v / (uint32_t)rows; const int r = (int)(v - t * (uint32_t)rows); return BnPart{t, r, r}; }
    return BnPart{hs + rs_tile_of_block(v - hs * (uint32_t)rows, nb - hs, true), 0, 0x7fffffff};
}

template <typename F>
__device__ __forceinline__ void bn_group_pairs(uint2 v, int shift, const GsrShard& sh, int stiles_x, int row_lo, int row_hi, F&& fn)
{
    const int lane = threadIdx.x & 63;
    const uint32_t rc = v.y;
    const int x0 = rc & 255, y0 = (rc >> 8) & 255, x1 = rc >> 16 & 255, y1 = rc >> 24;
    const bool some = x1 >= x0 && y1 >= y0;
    const int ya = max(y0 >> shift, row_lo), yb = min(y1 >> shift, row_hi);            // (the rows of the rect this work item owns)
    const int area = (some && yb >= ya) ? ((x1 >> shift) - (x0 >> shift) + 1) * (yb - ya + 1) : 0;
    // Which splats are "big"?  One covering more than BN_BIG super-tiles stalls its 63 neighbours if it walks its rect alone -- but the
    // wave expands big ones ONE AFTER THE OTHER, a dozen lanes busy each time, and a group where most splats are a little over
    // the threshold (a capture's background: every splat of a far wall covers 3 x 3 or 4 x 4 super-tiles, and depth order puts
    // them all next to each other) took 40 such rounds instead of 16 lock-step iterations: k_bin_place 176 us on R1 for 3.5 M pairs
    // against 55 us on C4 for 8 M.  So: where more than BN_MANY lanes of the group are over BN_BIG, the lanes walk rects of up to
    // BN_BIG_MANY super-tiles themselves, and only the really large ones are spread over the wave.  (Wave-uniform; k_bin_count and
    // both phases of k_bin_place see the same groups, so they agree -- and the pairs a group yields do not depend on who walks them.)
#ifndef BN_MANY
#define BN_MANY 6
#endif
#ifndef BN_BIG_MANY
#define BN_BIG_MANY 32
#endif
    const int thr = __builtin_popcountll(__ballot(area > BN_BIG)) > BN_MANY ? BN_BIG_MANY : BN_BIG;
    const bool big = area > thr;
    if (!big) bn_for_each_super(rc, shift, sh, stiles_x, row_lo, row_hi, [&](uint32_t d, int sx, int sy) { fn(lane, v, d, sx, sy); });
    unsigned long long bigs = __ballot(big);
    while (bigs) {
        const int L = __builtin_ctzll(bigs);
        bigs &= bigs - 1ull;
        const uint2 vL = make_uint2((uint32_t)__shfl((int)v.x, L, 64), (uint32_t)__shfl((int)v.y, L, 64));
        const uint32_t r = vL.y;
        const int X0 = r & 255, Y0 = (r >> 8) & 255, X1 = (r >> 16) & 255, Y1 = r >> 24;
        const int sx0 = X0 >> shift, sy0 = max(Y0 >> shift, row_lo);
        const int w = (X1 >> shift) - sx0 + 1, h = min(Y1 >> shift, row_hi) - sy0 + 1;
        auto cell = [&](int ry, int cx) __attribute__((always_inline)) {
            const int sy = sy0 + ry, sx = sx0 + cx;
            if (sh.count > 1) {
                const int lo = max(Y0, sy << shift), hi = min(Y1, ((sy + 1) << shift) - 1);
                if (gsr_owned_rect_rows(lo, hi, sh) == 0) return;
            }
            const uint32_t dd = (uint32_t)sy * (uint32_t)stiles_x + (uint32_t)sx;
            fn(L, vL, dd, sx, sy);
        };
        if (w <= 64) {
            // lanes as rows x columns of the rect, the columns padded to a power of two: no integer division (~30 instructions
            // here, and a capture's screen-filling background splats -- the LAST few hundred of the depth order, all in one
            // workgroup -- are expanded by one wave, one after the other: that workgroup is what the kernel waits for)
            const int cb = w > 1 ? 32 - __builtin_clz((unsigned)(w - 1)) : 0;     // 1 << cb = columns per row of lanes >= w
            const int cx = lane & ((1 << cb) - 1), rstep = 64 >> cb;
            if (cx < w)
                for (int ry = lane >> cb; ry < h; ry += rstep) cell(ry, cx);
        } else {   // (a frame more than 64 super-tiles wide: 16384 x 16 pixels)
            for (int t = lane; t < w * h; t += 64) { const int ry = t / w; cell(ry, t - ry * w); }
        }
    }
}

// sorted = (splat index, packed tile rect) in depth-rank order; *n_dev of them exist (the grid is
// sized for the host-side upper bound: surplus blocks publish zeros).  Blocks are handed to XCDs
// in contiguous eighths (rs_tile_of_block), like the depth sort's.
// ITEMS = splats per thread: 4 for frames that keep millions (fewer, larger workgroups: less histogram to scan), 1 or 2 for the
// few hundred thousand splats of an occlusion-culled frame (300 workgroups of 1024 splats leave most of the chip idle and every
// workgroup walks four groups one after the other)
template <int ITEMS>
__global__ void __launch_bounds__(BN_THREADS)
k_bin_count(const uint2* __restrict__ sorted, const uint32_t* __restrict__ n_dev, int shift, GsrShard sh,
            int stiles_x, int stiles_y, uint32_t* __restrict__ hist, uint32_t nblk)
{
    constexpr uint32_t TILE = BN_THREADS * ITEMS;
    // (d < BN_BINS below: a frame whose small-frame sort overflowed a bucket is rendered again, but until then its payloads may
    //  be another frame's -- they must not index past the bins)
    __shared__ uint32_t h[4][BN_BINS];
    KPROF_BLK_BEGIN
    const int wave = threadIdx.x >> 6;
    const uint32_t n = *n_dev;
    const uint32_t nb = (n + TILE - 1) / TILE;
    // the grid follows what the slot's previous frame kept (+25 %): a frame that keeps more simply loops (nblk = the row
    // length of hist, the host's upper bound on the blocks)
    const uint32_t nv = bn_items(nb, stiles_y);
    for (uint32_t wi = blockIdx.x; wi < nv; wi += gridDim.x) {
        for (int b = threadIdx.x; b < 4 * BN_BINS; b += BN_THREADS) (&h[0][0])[b] = 0;
        __syncthreads();
        const BnPart part = bn_part(wi, nb, stiles_y);
        const uint32_t tile = part.tile;
        {
            const uint32_t base = tile * TILE;
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                const uint32_t i = base + k * BN_THREADS + threadIdx.x;   // (every wave sees 64 consecutive splats)
                const uint2 v = (i < n) ? sorted[i] : make_uint2(0u, GSR_RECT_EMPTY);
                bn_group_pairs(v, shift, sh, stiles_x, part.row_lo, part.row_hi,
                               [&](int, uint2, uint32_t d, int, int) { if (d < (uint32_t)BN_BINS) atomicAdd(&h[wave][d], 1u); });
            }
        }
        __syncthreads();
        // (a part writes the columns of ITS rows of super-tiles only: the others belong to the block's other parts)
        // (... and the part that owns the LAST row also the columns behind the grid, up to BN_BINS: k_scan_rows scans all BN_BINS rows of
        //  `hist`, which the radix passes share -- stale counts there would end up in totals[n_super ..] and `offs`)
        const int b_lo = part.row_lo * stiles_x, b_hi = part.row_hi >= stiles_y - 1 ? BN_BINS : (part.row_hi + 1) * stiles_x;
        for (int b = b_lo + (int)threadIdx.x; b < b_hi; b += BN_THREADS)
            hist[(size_t)b * nblk + tile] = h[0][b] + h[1][b] + h[2][b] + h[3][b];
        __syncthreads();
    }
    KPROF_BLK_END(3, n < TILE ? n : TILE)
}

// list range of every super-tile = exclusive scan of the totals; the pair count and the frame's other news go to the host.
// The count is formed in 64 bits as well: past max_pairs (list positions are int32) every range is left empty and
// the host is told 0xffffffff -- it reports GSR_E_TOO_MANY_PAIRS instead of compositing wrapped positions.
struct GsrRangeArgs {
    const uint32_t* totals;            // [256] pairs per super-tile (k_scan_rows)
    int32_t n_super;
    int32_t *sstart, *send;            // out: list ranges
    volatile unsigned long long* host_total;   // pinned, mapped [4]: ticket << 32 | pair count; hints; surviving clusters; key range
    uint32_t ticket;
    unsigned long long max_pairs;
    uint32_t* redo_count;              // the frame's list of tiles given up by the plain blend kernel starts empty
    const uint32_t* lazy_hint;         // k_sum_work's verdict on the previous frame, forwarded to the host
    const uint32_t* n_sorted;          // splats that reached the depth sort (what the frame kept)
    const uint32_t* k1_counts;         // [1] = clusters that survived k_cluster_cull, [2] = the small-frame sort gave a bucket up
    const uint32_t* sorted_keys;       // the frame's keys in depth order
    const uint32_t* depth_active;      // depth-tested frames: "the depth buffer holds something in front of the far plane" (k_cluster.h), or NULL
};
// all 256 threads of a workgroup; returns this thread's (= super-tile's) list start.  publish: also write the ranges and the mailbox
__device__ __forceinline__ uint32_t bn_ranges(const GsrRangeArgs& a, bool publish, uint32_t* s_wave /*[4]*/, unsigned long long* s_sum /*[4]*/)
{
    if (publish && threadIdx.x == 0) {
        // word 1 of the mailbox: the hints in the low half, the frame's splat count in the high half; word 2: surviving clusters;
        // word 3: the smallest and the largest key of the frame (predicts the next frame's sort buckets, k_sort.h)
        *a.redo_count = 0u;
        const uint32_t ns = *a.n_sorted;
        // (bit 5 of the hints: the small-frame sort gave a bucket up -- the lists of this frame are not to be trusted)
        a.host_total[1] = ((unsigned long long)ns << 32) | (unsigned long long)(*a.lazy_hint & 31u) | (a.k1_counts[2] ? 32ull : 0ull) |
                          ((a.depth_active && *a.depth_active) ? 64ull : 0ull);
        a.host_total[2] = (unsigned long long)a.k1_counts[1];
        a.host_total[3] = ns ? ((unsigned long long)a.sorted_keys[ns - 1u] << 32) | (unsigned long long)a.sorted_keys[0] : 0ull;
    }
    const uint32_t v = ((int)threadIdx.x < a.n_super) ? a.totals[threadIdx.x] : 0u;
    unsigned long long w = v;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) w += __shfl_down(w, d, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = w;
    uint32_t tot;
    const uint32_t ex = block_excl_scan_256(v, s_wave, &tot);   // (its barriers publish s_sum too)
    const bool too_many = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3] > a.max_pairs;
    if (publish) {
        if ((int)threadIdx.x < a.n_super) {
            a.sstart[threadIdx.x] = too_many ? 0 : (int32_t)ex;
            a.send[threadIdx.x] = too_many ? 0 : (int32_t)(ex + v);
        }
        // the host sizes the list buffer from it; it recognises THIS frame's count by the ticket in the upper half
        if (threadIdx.x == 0) { *a.host_total = ((unsigned long long)a.ticket << 32) | (too_many ? 0xffffffffull : (unsigned long long)tot); __threadfence_system(); }
    }
    return too_many ? 0u : ex;
}
// on its own: only a frame whose list buffer does not exist yet (the host has to size it before anything is placed)
__global__ void __launch_bounds__(BN_BINS)
k_bin_ranges(GsrRangeArgs a)
{
    __shared__ uint32_t s_wave[4];
    __shared__ unsigned long long s_sum[4];
    (void)bn_ranges(a, true, s_wave, s_sum);
}

// Placement.  Inside a list the order must be the depth order of the splats, so the pairs of one
// super-tile are ranked by splat position: across blocks by the scanned histogram, and inside a block through 64-bit LANE
// MASKS in LDS -- for every GROUP of 64 consecutive splats (lane = splat) and every super-tile, the set of lanes whose splat
// covers it.  Every lane ORs its bit into the masks of the super-tiles it covers (A); thread d walks super-tile d's masks
// group by group and leaves, per group, the list position of the group's first pair (S); a pair's position is then that +
// the number of lower lanes in its own group's mask (B).  Nothing is mutated between A and B.
// A block's 4 * ITEMS groups are dealt to the waves ROUND ROBIN (wave w: groups w, w + 4, ...), as in k_bin_count: the
// screen-filling background splats of a capture are the LAST few hundred of the depth order -- with a contiguous quarter of
// the block per wave (round 4) one wave expanded all of them, and the kernel waited for it (R1: 176 us for 3.5 M pairs against
// 55 us for C4's 8 M).
// Dynamic LDS: lmask[4 * ITEMS groups][ns] (u64) followed by gpre[4 * ITEMS][ns] (u32), ns = n_super.
// ranges.totals != NULL: the list ranges are formed HERE (every workgroup scans the 256 totals itself; the last one also writes
// them out and posts the pair count) instead of by a k_bin_ranges launch in front: one launch floor less per frame.
template <int ITEMS, bool ZQ>
__global__ void __launch_bounds__(BN_THREADS)
k_bin_place(const uint2* __restrict__ sorted, const uint32_t* __restrict__ n_dev, int shift, GsrShard sh,
            int stiles_x, int ns, const uint32_t* __restrict__ offs, const int32_t* __restrict__ sstart,
            uint32_t nblk, uint32_t cap, uint2* __restrict__ out, GsrRangeArgs ranges,
            const float* __restrict__ zwin, float zq0, float zqs)
{
    static_assert(BN_THREADS == BN_BINS, "one thread per super-tile in the range scan");
    constexpr uint32_t TILE = BN_THREADS * ITEMS;
    constexpr int NG = 4 * ITEMS;                     // groups of 64 splats per block
    KPROF(0, 0)
    KPROF_BLK_BEGIN
    extern __shared__ unsigned long long bn_lds[];
    __shared__ uint32_t s_start[BN_BINS];
    __shared__ uint32_t s_rwave[4];
    __shared__ unsigned long long s_rsum[4];
    // (the grid is sized for the host's upper bound on the splats a frame keeps: most of its workgroups have nothing to do, and
    //  they leave before the range scan -- eleven thousand of them doing it first doubled the kernel on a culled C4 frame)
    const uint32_t n = *n_dev;
    const uint32_t nb = (n + TILE - 1) / TILE;
    // The LAST workgroup of the grid only publishes the ranges and the mailbox (four dependent trips: done by a workgroup with
    // splats of its own they ended the kernel three microseconds late); the others stride over the blocks -- the grid follows
    // what the slot's previous frame kept (+25 %) -- and one without a block leaves before the range scan.
    const bool publisher = ranges.totals && blockIdx.x == gridDim.x - 1u;
    const uint32_t workers = ranges.totals ? gridDim.x - 1u : gridDim.x;
    if (!publisher && blockIdx.x >= bn_items(nb, stiles_x > 0 ? (ns + stiles_x - 1) / stiles_x : 1)) return;
    if (ranges.totals) {
        s_start[threadIdx.x] = bn_ranges(ranges, publisher, s_rwave, s_rsum);
        if (publisher) return;
    } else {
        s_start[threadIdx.x] = ((int)threadIdx.x < ns) ? (uint32_t)sstart[threadIdx.x] : 0u;
    }
    __syncthreads();
    KPROF(0, 1)
    const int wave = threadIdx.x >> 6;
    unsigned long long* lmask = bn_lds;                                              // [group][d]
    uint32_t* gpre = reinterpret_cast<uint32_t*>(bn_lds + (size_t)NG * ns);          // [group][d]: list position of the group's first pair
    const int stiles_y = stiles_x > 0 ? (ns + stiles_x - 1) / stiles_x : 1;
    const uint32_t nv = bn_items(nb, stiles_y);
    for (uint32_t wi = blockIdx.x; wi < nv; wi += workers) {
    const BnPart part = bn_part(wi, nb, stiles_y);
    const uint32_t tile = part.tile;
    const uint32_t first = tile * TILE;
    KPROF(0, 2)
    for (int b = threadIdx.x; b < NG * ns; b += BN_THREADS) lmask[b] = 0ull;
    __syncthreads();
    KPROF(0, 3)
    uint2 v[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {   // (A) wave w: groups w, w + 4, ...
        const int grp = k * 4 + wave;
        const uint32_t i = first + (uint32_t)grp * 64u + (threadIdx.x & 63u);
        v[k] = (i < n) ? sorted[i] : make_uint2(0u, GSR_RECT_EMPTY);
        // (depth-tested frames: the coarse window depth goes into the index word's spare bits -- GsrFrame.idx_mask; the load is not needed before (B))
        if (ZQ && i < n) v[k].x |= gsr_zq(zwin[v[k].x], zq0, zqs) << GSR_ZQ_SHIFT;
        bn_group_pairs(v[k], shift, sh, stiles_x, part.row_lo, part.row_hi,
                       [&](int L, uint2, uint32_t d, int, int) { if (d < (uint32_t)ns) atomicOr(&lmask[grp * ns + d], 1ull << L); });
    }
    __syncthreads();
    KPROF(0, 4)
    for (int d = threadIdx.x; d < ns; d += BN_THREADS) {   // (S) per super-tile: where each group's pairs begin
        uint32_t p = s_start[d] + offs[(size_t)d * nblk + tile];
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) { gpre[grp * ns + d] = p; p += (uint32_t)__builtin_popcountll(lmask[grp * ns + d]); }
    }
    __syncthreads();
    KPROF(0, 6)
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {   // (B)
        const int grp = k * 4 + wave;
        bn_group_pairs(v[k], shift, sh, stiles_x, part.row_lo, part.row_hi, [&](int L, uint2 vL, uint32_t d, int sx, int sy) {
            if (d >= (uint32_t)ns) return;
            const uint32_t pos = gpre[grp * ns + d] + (uint32_t)__builtin_popcountll(lmask[grp * ns + d] & ((1ull << L) - 1ull));
            if (pos < cap) out[pos] = make_uint2(vL.x, bn_tile_mask(vL.y, sx, sy, shift));
        });
    }
    KPROF(0, 7)
    __syncthreads();   // (the next block re-uses the masks and the group positions)
    }
    KPROF_BLK_END(0, n < TILE ? n : TILE)
}

// root side of the multi-GPU path: gathered band images -> frame.
// gathered = count bands of band_rows x width pixels; band g holds rank g's tile rows stacked bottom-up
// (interleaved layout: rows g, g+count, ...; band layout, rpb > 0: rows g*rpb ...).
__global__ void __launch_bounds__(256)
k_stitch_bands(const float4* __restrict__ gathered, int count, int rpb, int band_rows, int width, int height,
               float4* __restrict__ out)
{
    const size_t p = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (p >= (size_t)width * height) return;
    const int x = (int)(p % width), y = (int)(p / width);
    const int trow = y >> 4;
    const int g = rpb > 0 ? trow / rpb : trow % count, lrow = rpb > 0 ? trow - g * rpb : trow / count;
    const int by = lrow * GSR_TILE_PX + (y & 15);
    out[p] = gathered[((size_t)g * band_rows + by) * width + x];
}

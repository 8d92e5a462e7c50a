// k_cluster.h -- spatially ordered storage and cluster culling in front of K1.
//
// The reference streams every splat through its vertex shader every frame
// (/root/reference/gsplat_plugin/src/GSplatRenderer.C:647: one instanced quad per splat) and sorts all of them on the
// CPU (:176-216).  Here the splats are stored in MORTON ORDER of their positions (an upload-time permutation; ties in the
// depth sort are broken by this storage order -- the reference's own tie order is unspecified, :206-207), so that 64
// consecutive splats = one CLUSTER = one wavefront of K1 occupy a small box in space, and every frame starts with
//   k_cluster_cull   one thread per cluster: the cluster's box (AABB of the positions + the largest splat extent in it)
//                    against the clip planes, the frame's screen / this rank's band of tile rows, and the depth horizons
//                    of the tiles its screen bound reaches -> the ordered list of the clusters that may draw something.
// K1 and everything behind it run over the survivors only.  Every test is CONSERVATIVE with respect to the per-splat
// rules of k_preprocess.h (a cluster is dropped only if each of its splats would have been dropped there, or would have
// left no list entry), so frames are bit-identical with the stage switched off (GSR_OPT_CLUSTER_CULL = 0).
#pragma once
#include "gsr_device.h"

#define GSR_CLUSTER 64                 // splats per cluster = lanes of a wavefront
#ifndef CC_THREADS
#define CC_THREADS 256
#endif
// (CC_THREADS: threads of a k_cluster_cull workgroup, one cluster each per round)
#define CC_MAX_GROUPS 4096             // count entries K1's prologue can search (64 x 64)

// ---- upload time ------------------------------------------------------------------------------------------------------
// 30-bit Morton code of the position inside the cloud's bounding box (10 bits per axis); non-finite positions sort last
__device__ __forceinline__ uint32_t cc_spread10(uint32_t v)
{
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ void __launch_bounds__(256)
k_morton_codes(const float* __restrict__ P /* raw float[3] positions, upload order */, uint32_t n, float lx, float ly, float lz, float sx, float sy, float sz,
               uint32_t* __restrict__ code, uint32_t* __restrict__ idx)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float3 a = make_float3(P[3 * (size_t)i], P[3 * (size_t)i + 1], P[3 * (size_t)i + 2]);
    uint32_t c = 0x3fffffffu;
    if (__builtin_fabsf(a.x) < 3.0e38f && __builtin_fabsf(a.y) < 3.0e38f && __builtin_fabsf(a.z) < 3.0e38f) {
        const float qx = __builtin_fminf(__builtin_fmaxf((a.x - lx) * sx, 0.0f), 1023.0f);
        const float qy = __builtin_fminf(__builtin_fmaxf((a.y - ly) * sy, 0.0f), 1023.0f);
        const float qz = __builtin_fminf(__builtin_fmaxf((a.z - lz) * sz, 0.0f), 1023.0f);
        c = cc_spread10((uint32_t)qx) | (cc_spread10((uint32_t)qy) << 1) | (cc_spread10((uint32_t)qz) << 2);
    }
    code[i] = c;
    idx[i] = i;
}

// (the splats are written into storage order, and every 64 consecutive slots = one cluster get their bounds -- clusA = (lo.xyz of the
//  positions, the largest |diag(scale) R^T|_F bound), clusB = (hi.xyz, flag); flag != 0: a position or an extent bound is not finite, the
//  cluster is never culled -- by k_pack, k_preprocess.h)

// ---- per frame ----------------------------------------------------------------------------------------------------------
// Depth horizons live in a 6-level max pyramid over the tile grid (k_blend.h: k_tile_pass forms the per-tile horizons,
// k_horizon_dilate widens each by the frame's dilation radius and builds the levels): level l cell (x, y) = the largest
// (dilated) horizon of the tiles [x 2^l, (x+1) 2^l) x [y 2^l, (y+1) 2^l); +inf = no horizon (nothing may be culled there), 0 =
// a tile of another rank (nothing is needed there).
#define GSR_PYR_LEVELS 6
#define GSR_MAX_TILES_SIDE 1024        // GSR_MAX_DIM / GSR_TILE_PX
#define GSR_PYR_FLOATS (1024 * 1024 + 512 * 512 + 256 * 256 + 128 * 128 + 64 * 64 + 32 * 32 + 16)   // a grid of up to 1024 x 1024 tiles
#define GSR_DILATE_EXACT_MAX 3         // the dilation k_horizon_dilate applies tile by tile; what a frame wants beyond that
                                       // is added by widening the rects at look-up time
__host__ __device__ __forceinline__ int gsr_pyr_dim(int tiles, int level) { return ((tiles - 1) >> level) + 1; }
// largest horizon over the tile rect [x0, x1] x [y0, y1] (inside the grid), widened to the 2 x 2 cells of the finest level at
// which it spans no more than that -- or, beyond the top level, to up to 4 x 4 of its cells; +inf for a still larger rect.
// MONOTONE: a rect that contains another never gets a smaller value (its cells are unions of the other's), which is what lets
// k_tile_pass check a tile against the value of the tile itself -- every splat that touches the tile was compared with at
// least that.
__device__ __forceinline__ float gsr_pyr_max(const float* __restrict__ pyr, const int32_t* pyr_off, int tiles_x, int x0, int y0, int x1, int y1)
{
    const int span = max(x1 - x0, y1 - y0);
    int L = 31 - __builtin_clz((uint32_t)span | 1u);
    if (((x1 >> L) - (x0 >> L)) > 1 || ((y1 >> L) - (y0 >> L)) > 1) ++L;
    if (L < GSR_PYR_LEVELS) {
        const int w = gsr_pyr_dim(tiles_x, L);
        const float* p = pyr + pyr_off[L];
        const int a0 = x0 >> L, a1 = x1 >> L, b0 = y0 >> L, b1 = y1 >> L;
        return __builtin_fmaxf(__builtin_fmaxf(p[b0 * w + a0], p[b0 * w + a1]), __builtin_fmaxf(p[b1 * w + a0], p[b1 * w + a1]));
    }
    const int T = GSR_PYR_LEVELS - 1;
    const int a0 = x0 >> T, a1 = x1 >> T, b0 = y0 >> T, b1 = y1 >> T;
    if (a1 - a0 > 3 || b1 - b0 > 3) return __builtin_inff();
    const int w = gsr_pyr_dim(tiles_x, T);
    const float* p = pyr + pyr_off[T];
    float h = 0.0f;
    for (int b = b0; b <= b1; ++b)
        for (int a = a0; a <= a1; ++a) h = __builtin_fmaxf(h, p[b * w + a]);
    return h;
}

// ---- depth-tested frames (the reference draws with the depth test on, src/GSplatRenderer.C:595-610) -------------------
// What the opaque pass left, as TWO tile-max pyramids over the same tile grid and level offsets as the horizons (levels 0..3):
//   pyr   the largest depth under the tile.  A splat whose window depth exceeds it for every tile its rect reaches fails the test
//         `zwin <= depth[pixel]` at every fragment: K1 and k_cluster_cull drop it (no colour, no record, no sorting, no list entry)
//         instead of k_blend discarding it fragment by fragment.  Exact, whatever else the frame does.
//   pyrc  the largest depth among the tile's COVERED pixels (depth < 1: the opaque pass drew something there), 0 if it has none.
//         A covered pixel may never saturate, so no depth horizon can speak for it: the horizons are formed from the tiles'
//         uncovered pixels alone (k_blend.h: bit 15 of the bookkeeping), and a splat beyond the horizons of its rect is dropped
//         only if it is ALSO behind everything the opaque pass left under the covered pixels there (zwin > pyrc): the covered pixels
//         get what lies in front of the geometry, exactly, without any prediction.
// Values are clamped to >= 0 (window depths are >= 0, and a NaN pixel passes nothing), so their bit patterns order like unsigned
// integers.  active[par] != 0 iff some pixel is covered: every window depth K1 keeps is <= 1, so under a buffer that was merely
// cleared to the far plane nothing changes and the look-ups are skipped.
struct GsrDepthCull {
    const float* pyr;          // NULL = no culling against the opaque pass's depth
    const float* pyrc;
    const uint32_t* active;    // the word of this frame's parity
};
struct GsrDepthPyrArgs {
    const float* depth;        // [height][width] window depth of the opaque pass (row 0 = bottom, like the framebuffer)
    float* pyr;                // the two pyramids of this frame's parity ...
    float* pyrc;
    float* tcov;               // [tiles_y][tiles_x] for k_blend: < 0 = the tile has NO covered pixel (an ordinary tile: no depth load there)
    const float* stat;         // [tiles_y][tiles_x] or NULL: != 0 = the tile was CLASSIC in the frame that left this frame's horizons (k_blend.h:
                               // GsrHorizonArgs.stat) -- all of its pixels are expected to saturate inside its horizon, and no depth clause
                               // is applied on its account: its covered depths do not enter pyrc
    float* pyr_next;           // ... and the other parity's: levels 4 and 5 are max-reduced with atomics (like the horizons'), so this frame
    float* pyrc_next;          //     clears them for the slot's next depth-tested frame
    uint32_t* active;          // [2]: this frame sets [par], and clears [par ^ 1] likewise
    int32_t par;
    int32_t width, height, tiles_x, tiles_y;
    int32_t off[GSR_PYR_LEVELS];
};
// largest value over the tile rect [x0, x1] x [y0, y1] of a depth pyramid (levels 0 .. GSR_DPYR_LEVELS - 1); +inf -- "unknown": nothing
// is culled on its account -- for a rect that spans more than 2 x 2 cells of the top level (more than 512 pixels)
#define GSR_DPYR_LEVELS 5
__device__ __forceinline__ float gsr_dpyr_max(const float* __restrict__ pyr, const int32_t* pyr_off, int tiles_x, int x0, int y0, int x1, int y1)
{
    const int span = max(x1 - x0, y1 - y0);
    int L = 31 - __builtin_clz((uint32_t)span | 1u);
    if (((x1 >> L) - (x0 >> L)) > 1 || ((y1 >> L) - (y0 >> L)) > 1) ++L;
    if (L >= GSR_DPYR_LEVELS) return __builtin_inff();
    const int w = gsr_pyr_dim(tiles_x, L);
    const float* p = pyr + pyr_off[L];
    const int a0 = x0 >> L, a1 = x1 >> L, b0 = y0 >> L, b1 = y1 >> L;
    return __builtin_fmaxf(__builtin_fmaxf(p[b0 * w + a0], p[b0 * w + a1]), __builtin_fmaxf(p[b1 * w + a0], p[b1 * w + a1]));
}
// One workgroup per 8 x 8 block of tiles (128 x 128 pixels, 64 KB), THREADS = 1024 in the launch of its own (four 16-byte loads per
// thread, issued together: the pass is a latency chain, not a bandwidth problem -- 8.3 MB at 1080p), 256 as the first workgroups of
// k_cluster_cull's launch (sixteen loads per thread, hidden beside the cluster tests).  Tile maxima meet in LDS, the first wavefront
// (lane = tile in Morton order) writes level 0 and shuffles levels 1..3 together like k_horizon_dilate; level 4 is max-reduced with
// two atomics per workgroup into cells the slot's PREVIOUS depth-tested frame cleared (the pyramids are double-buffered by frame
// parity for that; atomics on one cache line serialise at ~12 ns each: one per 2 x 2-tile group on levels 2..5 took 10 us).
__host__ __device__ __forceinline__ int gsr_depth_pyramid_blocks(int tiles_x, int tiles_y) { return ((tiles_x + 7) >> 3) * ((tiles_y + 7) >> 3); }
template <int THREADS>
__device__ __forceinline__ void gsr_depth_pyramid_block(const GsrDepthPyrArgs& a, const int b)
{
    static_assert(THREADS == 256 || THREADS == 1024, "rows of 32 lanes");
    constexpr int RP = THREADS / 32, NL = 128 / RP;          // pixel rows per pass, loads per thread
    __shared__ uint32_t s_tile[64], s_tilec[64], s_tcov[64], s_cov;
    const int tid = threadIdx.x;
    const int nbx = (a.tiles_x + 7) >> 3;
    const int by = b / nbx, bx = b - by * nbx;
    if (tid < 64) { s_tile[tid] = 0u; s_tilec[tid] = 0u; s_tcov[tid] = 0u; }
    if (tid == 0) s_cov = 0u;
    if (b == 0) {
        if (tid == 0) a.active[a.par ^ 1] = 0u;
        const int n4 = gsr_pyr_dim(a.tiles_x, 4) * gsr_pyr_dim(a.tiles_y, 4);
        for (int i = tid; i < n4; i += THREADS) { a.pyr_next[a.off[4] + i] = 0.0f; a.pyrc_next[a.off[4] + i] = 0.0f; }
    }
    __syncthreads();
    const int c4 = tid & 31, r = tid >> 5;
    const int px0 = bx * 128 + c4 * 4;
    const bool vec = ((a.width & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.depth) & 15u) == 0u);   // (uniform)
    float4 q[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {            // (all loads first)
        const int row = by * 128 + k * RP + r;
        q[k] = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);   // (no pixel: neither a depth nor a covered one)
        if (row < a.height && px0 < a.width) {
            const float* src = a.depth + (size_t)row * a.width + px0;
            if (vec) {
                q[k] = *reinterpret_cast<const float4*>(src);
            } else {
                q[k].x = src[0];
                if (px0 + 1 < a.width) q[k].y = src[1];
                if (px0 + 2 < a.width) q[k].z = src[2];
                if (px0 + 3 < a.width) q[k].w = src[3];
            }
        }
    }
    // (per pixel: fmax drops a NaN operand, so a NaN pixel counts as depth 0 -- it passes nothing; covered = depth < 1, which a NaN is not)
    auto dep = [](float d) { return __builtin_fmaxf(d, 0.0f); };
    auto cov = [](float d) { return (d < 1.0f && d >= 0.0f) ? d : 0.0f; };    // (a negative depth passes nothing either: 0)
    auto iscov = [](float d) { return !(d >= 1.0f); };                        // (a NaN pixel counts as covered: it is not "cleared")
    bool anyc = false;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int rl = k * RP + r, row = by * 128 + rl;
        const bool e0 = row < a.height && px0 < a.width, e1 = e0 && px0 + 1 < a.width, e2 = e0 && px0 + 2 < a.width, e3 = e0 && px0 + 3 < a.width;
        float v = __builtin_fmaxf(__builtin_fmaxf(dep(q[k].x), dep(q[k].y)), __builtin_fmaxf(dep(q[k].z), dep(q[k].w)));
        float vc = __builtin_fmaxf(__builtin_fmaxf(cov(q[k].x), cov(q[k].y)), __builtin_fmaxf(cov(q[k].z), cov(q[k].w)));
        const bool cv = (e0 && iscov(q[k].x)) || (e1 && iscov(q[k].y)) || (e2 && iscov(q[k].z)) || (e3 && iscov(q[k].w));
        anyc = anyc || cv;
        v = __builtin_fmaxf(v, __shfl_xor(v, 1, 64)); vc = __builtin_fmaxf(vc, __shfl_xor(vc, 1, 64));
        v = __builtin_fmaxf(v, __shfl_xor(v, 2, 64)); vc = __builtin_fmaxf(vc, __shfl_xor(vc, 2, 64));
        const int t = rl >> 4;                // tile row of the block
        if ((c4 & 3) == 0 && v > 0.0f) atomicMax(&s_tile[t * 8 + (c4 >> 2)], __float_as_uint(v));
        if ((c4 & 3) == 0 && vc > 0.0f) atomicMax(&s_tilec[t * 8 + (c4 >> 2)], __float_as_uint(vc));
        if (cv) s_tcov[t * 8 + (c4 >> 2)] = 1u;         // (plain stores of the same value)
    }
    if (__any(anyc) && (tid & 63) == 0) s_cov = 1u;
    __syncthreads();
    if (tid >= 64) return;
    const int lane = tid;
    const int lx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), ly = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int tx = bx * 8 + lx, ty = by * 8 + ly;
    const bool inside = tx < a.tiles_x && ty < a.tiles_y;
    float v = inside ? __uint_as_float(s_tile[ly * 8 + lx]) : 0.0f, vc = inside ? __uint_as_float(s_tilec[ly * 8 + lx]) : 0.0f;
    if (inside) {
        const bool covt = s_tcov[ly * 8 + lx] != 0u;
        a.tcov[ty * a.tiles_x + tx] = covt ? vc : -1.0f;
        if (!covt || (a.stat && a.stat[ty * a.tiles_x + tx] != 0.0f)) vc = 0.0f;      // (no covered pixel / no depth clause for a classic tile)
        a.pyr[a.off[0] + ty * a.tiles_x + tx] = v; a.pyrc[a.off[0] + ty * a.tiles_x + tx] = vc;
    }
    if (s_cov && lane == 0) atomicOr(&a.active[a.par], 1u);
    v = __builtin_fmaxf(v, __shfl_xor(v, 1, 64)); v = __builtin_fmaxf(v, __shfl_xor(v, 2, 64));
    vc = __builtin_fmaxf(vc, __shfl_xor(vc, 1, 64)); vc = __builtin_fmaxf(vc, __shfl_xor(vc, 2, 64));
    if ((lane & 3) == 0 && inside) { const int o = a.off[1] + (ty >> 1) * gsr_pyr_dim(a.tiles_x, 1) + (tx >> 1); a.pyr[o] = v; a.pyrc[o] = vc; }
    v = __builtin_fmaxf(v, __shfl_xor(v, 4, 64)); v = __builtin_fmaxf(v, __shfl_xor(v, 8, 64));
    vc = __builtin_fmaxf(vc, __shfl_xor(vc, 4, 64)); vc = __builtin_fmaxf(vc, __shfl_xor(vc, 8, 64));
    if ((lane & 15) == 0 && inside) { const int o = a.off[2] + (ty >> 2) * gsr_pyr_dim(a.tiles_x, 2) + (tx >> 2); a.pyr[o] = v; a.pyrc[o] = vc; }
    v = __builtin_fmaxf(v, __shfl_xor(v, 16, 64)); v = __builtin_fmaxf(v, __shfl_xor(v, 32, 64));
    vc = __builtin_fmaxf(vc, __shfl_xor(vc, 16, 64)); vc = __builtin_fmaxf(vc, __shfl_xor(vc, 32, 64));
    if (lane == 0) {
        a.pyr[a.off[3] + by * nbx + bx] = v; a.pyrc[a.off[3] + by * nbx + bx] = vc;
        // (level 4: values are >= 0, their bit patterns order like unsigned integers)
        const int o4 = a.off[4] + (by >> 1) * gsr_pyr_dim(a.tiles_x, 4) + (bx >> 1);
        if (v > 0.0f) atomicMax(reinterpret_cast<uint32_t*>(a.pyr) + o4, __float_as_uint(v));
        if (vc > 0.0f) atomicMax(reinterpret_cast<uint32_t*>(a.pyrc) + o4, __float_as_uint(vc));
    }
}
// Frames whose previous depth buffer was CLEAR (the common case: a scene of splats only) do not build pyramids at all: if this buffer is
// clear too nobody will read them -- every look-up is behind `active` -- so all the frame needs is that word: is any pixel covered?
// 4096 pixels per workgroup of 256 threads, four 16-byte loads each, beside the cluster tests of the same launch (the pyramid blocks there
// -- sixteen loads per thread, LDS, shuffles -- made that launch 1.5 us longer than a plain frame's).  If geometry has APPEARED, the word
// says so, the frame goes on without depth culling (the depth-tested blend kernel compares every fragment anyway; a frame that was culled
// against horizons is rendered again: gsr_api.hip) and the slot's next frame builds the pyramids in a launch of its own.
__host__ __device__ __forceinline__ int gsr_depth_detect_blocks(int width, int height) { return (int)(((long long)width * height + 4095) / 4096); }
__device__ __forceinline__ void gsr_depth_detect_block(const GsrDepthPyrArgs& a, const int b)
{
    const int tid = threadIdx.x;
    if (b == 0) {      // (the parity hand-over of the full pass: gsr_depth_pyramid_block)
        if (tid == 0) a.active[a.par ^ 1] = 0u;
        const int n4 = gsr_pyr_dim(a.tiles_x, 4) * gsr_pyr_dim(a.tiles_y, 4);
        for (int i = tid; i < n4; i += 256) { a.pyr_next[a.off[4] + i] = 0.0f; a.pyrc_next[a.off[4] + i] = 0.0f; }
    }
    const long long n = (long long)a.width * a.height;
    const bool vec = (reinterpret_cast<uintptr_t>(a.depth) & 15u) == 0u;   // (uniform)
    float4 q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = ((long long)b * 1024 + k * 256 + tid) * 4;
        q[k] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        if (vec && i + 3 < n) q[k] = *reinterpret_cast<const float4*>(a.depth + i);
        else {
            if (i < n) q[k].x = a.depth[i];
            if (i + 1 < n) q[k].y = a.depth[i + 1];
            if (i + 2 < n) q[k].z = a.depth[i + 2];
            if (i + 3 < n) q[k].w = a.depth[i + 3];
        }
    }
    bool cov = false;          // (a NaN pixel counts as covered: it is not "cleared")
#pragma unroll
    for (int k = 0; k < 4; ++k) cov = cov || !(q[k].x >= 1.0f) || !(q[k].y >= 1.0f) || !(q[k].z >= 1.0f) || !(q[k].w >= 1.0f);
    if (__any(cov) && (tid & 63) == 0) atomicOr(&a.active[a.par], 1u);
}
// a launch of its own: frames whose previous depth buffer held opaque geometry (k_cluster_cull then culls against the pyramid too)
__global__ void __launch_bounds__(1024)
k_depth_pyramid(GsrDepthPyrArgs a) { gsr_depth_pyramid_block<1024>(a, (int)blockIdx.x); }

// Front-slab frames: the slab key = the end of the first histogram bin by which `want` of the surviving clusters have begun
// (want = clamp(total * frac_num / 256, min_clusters, max_clusters)); everything when the frame keeps fewer than twice that.
// Every workgroup of phase 1's second k_cluster_cull pass works it out for itself from the first pass's histogram (a scan of
// GSR_SLAB_BINS bins: cheaper than a launch in between); workgroup 0 also leaves it in memory for everything behind:
// slab[0] = the key (relative to key_min, inclusive), slab[1] = surviving clusters; slab[2], [3] = first key and bucket shift
// of the small-frame sort (k_sort.h) over phase 1's keys [0, key], slab[4], [5] = the same for phase 2's keys (key, key_range].
// The histogram is cleared by phase 2's pass.
struct GsrSlabPick {
    uint32_t min_clusters, max_clusters, frac_num, key_range;
};
__device__ __forceinline__ uint32_t gsr_slab_pick(const uint32_t* __restrict__ hist, int hist_shift, const GsrSlabPick& pk, uint32_t* __restrict__ slab,
                                                  uint32_t* s_wave /* [CC_THREADS / 64] */, uint32_t* s_pick /* [1] */)
{
    static_assert(GSR_SLAB_BINS % CC_THREADS == 0, "bins per thread");
    constexpr int PER = GSR_SLAB_BINS / CC_THREADS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { v[k] = hist[t * PER + k]; sum += v[k]; }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) s_wave[wave] = inc;
    if (t == 0) *s_pick = 0xffffffffu;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < CC_THREADS / 64; ++w) { const uint32_t c = s_wave[w]; before += w < wave ? c : 0u; total += c; }
    uint32_t want = (uint32_t)(((unsigned long long)total * pk.frac_num) >> 8);
    want = want > pk.max_clusters ? pk.max_clusters : want;
    want = want < pk.min_clusters ? pk.min_clusters : want;
    uint32_t run = before + inc - sum;                       // clusters in the bins before this thread's
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (total >= 2u * want && run < want && run + v[k] >= want) *s_pick = (uint32_t)(t * PER + k);   // the bin in which the count crosses `want`
        run += v[k];
    }
    __syncthreads();
    const uint32_t b = *s_pick;
    const unsigned long long end = ((unsigned long long)(b + 1u) << hist_shift) - 1ull;
    const uint32_t key = (b == 0xffffffffu || end > 0xfffffffeull || end >= (unsigned long long)pk.key_range) ? 0xffffffffu : (uint32_t)end;
    if (blockIdx.x == 0 && t == 0) {
        slab[0] = key;
        slab[1] = total;
        // BK_BUCKETS equal buckets over each phase's key range
        const unsigned long long w1 = (key == 0xffffffffu ? (unsigned long long)pk.key_range : (unsigned long long)key) + 1ull;
        int s1 = 0;
        while (s1 < 31 && (w1 >> s1) > (unsigned long long)BK_BUCKETS) ++s1;
        slab[2] = 0u; slab[3] = (uint32_t)s1;
        const unsigned long long w2 = key == 0xffffffffu ? 1ull : (unsigned long long)pk.key_range - key;
        int s2 = 0;
        while (s2 < 31 && (w2 >> s2) > (unsigned long long)BK_BUCKETS) ++s2;
        slab[4] = key == 0xffffffffu ? 0u : key + 1u; slab[5] = (uint32_t)s2;
    }
    return key;
}

// one thread per cluster; workgroup b handles the clusters [b * per, (b + 1) * per), per = CC_THREADS * rounds, and leaves
// the ones that stay, in cluster order, at seg[b * per ...] with their number in cnt[b]
// Front-slab frames (gsr_api.hip): phase 1 also leaves a HISTOGRAM of the surviving clusters' nearest sort keys (slab_hist:
// GSR_SLAB_BINS bins of width 2^hist_shift over the frame's key range; a cluster without a usable bound counts as nearest), from
// which k_slab_pick takes the slab key; phase 2 reads that key (slab_key) and drops the clusters that lie wholly in front of it.
// mode (front-slab frames): 0 = an ordinary frame; 1 = phase 1's first pass: the histogram (slab_hist is written; workgroup 0 also
// clears the top levels of the pyramid k_slab_mid is going to fill, zero_f[0, zero_n)); 2 = phase 1's second pass: the slab key
// from the histogram (gsr_slab_pick; slab_hist is read), only the slab's clusters stay; 3 = phase 2: slab[0] is read, the clusters
// wholly inside the slab go, the rest is culled against the tiles phase 1 finished (hpyr), and the histogram is cleared.
// (DEPTH = false: the instantiation frames without a depth buffer run -- no pyramid workgroups, no depth tests, none of their registers:
//  with everything in one kernel the plain frame's launch was 0.9 us longer than round 5's)
template <bool DEPTH>
__global__ void __launch_bounds__(CC_THREADS)
k_cluster_cull(GsrFrame f, const float4* __restrict__ clusA, const float4* __restrict__ clusB, uint32_t nclus, int rounds,
               int enabled, const float* __restrict__ hpyr /* or NULL: no occlusion test */,
               uint32_t* __restrict__ seg, uint32_t* __restrict__ cnt,
               int mode, uint32_t* __restrict__ slab_hist, int hist_shift, uint32_t* __restrict__ slab, GsrSlabPick pk,
               float* __restrict__ zero_f, int zero_n,
               uint32_t* __restrict__ bk_zero /* the small-frame sort's bucket counters (BK_BUCKETS, BK_STRIDE apart), or NULL */,
               uint32_t* __restrict__ flag_zero /* ... and its "gave a bucket up" flag: cleared here, BEFORE K1, whose workgroups scatter
                                                   into the buckets themselves */,
               GsrDepthPyrArgs dp, uint32_t n_dp /* depth-tested frames whose previous depth buffer was clear: the FIRST n_dp workgroups look
                                                    whether this one holds a covered pixel (gsr_depth_detect_block) beside the cluster tests */,
               GsrDepthCull dc /* ... or the pyramid exists already (k_depth_pyramid ran in front): clusters are tested against it */)
{
    static_assert(CC_THREADS == 256, "the folded depth pyramid workgroups are gsr_depth_pyramid_block<256>");
    if (DEPTH && blockIdx.x < n_dp) { gsr_depth_detect_block(dp, (int)blockIdx.x); return; }
    const uint32_t bid = DEPTH ? blockIdx.x - n_dp : blockIdx.x, nbid = DEPTH ? gridDim.x - n_dp : gridDim.x;
    const bool dact = DEPTH && dc.pyr != nullptr && *dc.active != 0u;     // (uniform)
    __shared__ uint32_t s_w[CC_THREADS / 64];
    __shared__ uint32_t s_hist[GSR_SLAB_BINS];
    __shared__ uint32_t s_pick;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t per = (uint32_t)CC_THREADS * (uint32_t)rounds;
    uint32_t kept = 0;
    uint32_t key_a = 0xffffffffu;
    if (mode == 1) {
        for (int b = threadIdx.x; b < GSR_SLAB_BINS; b += CC_THREADS) s_hist[b] = 0u;
        if (bid == 0) for (int i = threadIdx.x; i < zero_n; i += CC_THREADS) zero_f[i] = 0.0f;
        __syncthreads();
    } else if (mode == 2) {
        key_a = gsr_slab_pick(slab_hist, hist_shift, pk, slab, s_w, &s_pick);
        __syncthreads();                                       // (s_w is used again below)
    } else if (mode == 3) {
        key_a = slab[0];
        if (bid == 0) for (int b = threadIdx.x; b < GSR_SLAB_BINS; b += CC_THREADS) slab_hist[b] = 0u;
    }
    if (bid == nbid - 1u) {
        if (bk_zero) for (int d = threadIdx.x; d < BK_BUCKETS; d += CC_THREADS) bk_zero[(size_t)d * BK_STRIDE] = 0u;
        if (threadIdx.x == 0 && flag_zero) *flag_zero = 0u;
    }
    const bool want_hist = mode == 1;
    for (int r = 0; r < rounds; ++r) {
        const uint32_t cl = bid * per + (uint32_t)r * CC_THREADS + threadIdx.x;
        bool keep = cl < nclus;
        uint32_t kb_near = 0u;                 // a lower bound of the cluster's sort keys (0 = unknown)
        if (keep && enabled) {
            const float4 A = clusA[cl], B = clusB[cl];
            if (B.w == 0.0f) {
                // the box, widened by what the GSplatOrigin round trip (fl32(P - origin) + origin) can move a position
                const float plo[3] = {A.x, A.y, A.z}, phi[3] = {B.x, B.y, B.z};
                float lo[3] = {A.x, A.y, A.z}, hi[3] = {B.x, B.y, B.z};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float e = 1.0e-6f * (__builtin_fmaxf(__builtin_fabsf(lo[k]), __builtin_fabsf(hi[k])) + __builtin_fabsf(f.origin[k])) + 1.0e-30f;
                    lo[k] -= e; hi[k] += e;
                }
                // clip coordinates are affine in the position: their extrema over the box are at its corners
                float wmin = 3.0e38f, wmax = -3.0e38f, zpw_max = -3.0e38f, wmz_max = -3.0e38f, tz_min = 3.0e38f, tz_max = -3.0e38f;
                float cxmin = 3.0e38f, cxmax = -3.0e38f, cymin = 3.0e38f, cymax = -3.0e38f, mag = 0.0f, znmin = 3.0e38f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float x = (c & 1) ? hi[0] : lo[0], y = (c & 2) ? hi[1] : lo[1], z = (c & 4) ? hi[2] : lo[2];
                    const float tvx = gsr_fma(f.ov[0], x, gsr_fma(f.ov[1], y, gsr_fma(f.ov[2], z, f.ov[3])));
                    const float tvy = -gsr_fma(f.ov[4], x, gsr_fma(f.ov[5], y, gsr_fma(f.ov[6], z, f.ov[7])));
                    const float tvz = gsr_fma(f.ov[8], x, gsr_fma(f.ov[9], y, gsr_fma(f.ov[10], z, f.ov[11])));
                    const float clx = gsr_fma(f.pr[0], tvx, gsr_fma(f.pr[1], tvy, gsr_fma(f.pr[2], tvz, f.pr[3])));
                    const float cly = gsr_fma(f.pr[4], tvx, gsr_fma(f.pr[5], tvy, gsr_fma(f.pr[6], tvz, f.pr[7])));
                    const float clz = gsr_fma(f.pr[8], tvx, gsr_fma(f.pr[9], tvy, gsr_fma(f.pr[10], tvz, f.pr[11])));
                    const float clw = gsr_fma(f.pr[12], tvx, gsr_fma(f.pr[13], tvy, gsr_fma(f.pr[14], tvz, f.pr[15])));
                    const float tz = gsr_fma(f.vw[8], x, gsr_fma(f.vw[9], y, gsr_fma(f.vw[10], z, f.vw[11])));
                    mag = __builtin_fmaxf(mag, __builtin_fabsf(tvx) + __builtin_fabsf(tvy) + __builtin_fabsf(tvz) + 1.0f);
                    wmin = __builtin_fminf(wmin, clw); wmax = __builtin_fmaxf(wmax, clw);
                    zpw_max = __builtin_fmaxf(zpw_max, clz + clw);     // < 0 everywhere: in front of the near plane
                    wmz_max = __builtin_fmaxf(wmz_max, clw - clz);     // < 0 everywhere: beyond the far plane
                    tz_min = __builtin_fminf(tz_min, tz); tz_max = __builtin_fmaxf(tz_max, tz);
                    const float iw = 1.0f / clw;
                    const float px = gsr_fma(clx * iw, 0.5f, 0.5f) * f.W, py = gsr_fma((-cly) * iw, 0.5f, 0.5f) * f.H;
                    cxmin = __builtin_fminf(cxmin, px); cxmax = __builtin_fmaxf(cxmax, px);
                    cymin = __builtin_fminf(cymin, py); cymax = __builtin_fmaxf(cymax, py);
                    znmin = __builtin_fminf(znmin, clz * iw);          // (z / w is monotone along any line where w > 0: extrema at the corners too)
                }
                // rounding of the affine forms (here and in K1): a few ulp of the largest term
                float prn = 0.0f;
#pragma unroll
                for (int k = 0; k < 16; ++k) prn = __builtin_fmaxf(prn, __builtin_fabsf(f.pr[k]));
                const float eps = 1.0e-5f * mag * (prn + 1.0f);
                if (wmax < -eps || zpw_max < -eps || wmz_max < -eps) {
                    keep = false;                                       // w <= 0, or z outside [-w, w], for every splat
                } else if (wmin > eps && (tz_max < -1.0e-6f * mag || tz_min > 1.0e-6f * mag) && mag < 1.0e15f) {
                    // wholly in front of the eye: the screen positions of its splats lie inside the hull of the projected
                    // corners, and every quad inside its centre +- hb: the cheap extent bound of gsr_k1_front, taken at the
                    // largest |diag(scale) R^T|_F and the smallest |view z| of the cluster
                    // the bbox half extents of a quad are rq (s1 |ex| + s2 |ey|) <= 2 sqrt(s1^2 + s2^2) = 2 sqrt(2 (lambda1 + lambda2)) <=
                    // 2 sqrt(2 (trace(cov2d) + 0.6)), and trace(J W S W^T J^T) <= |J|_2^2 |W|_2^2 trace(S) with |J|_2^2 = (f/tz)^2 (1 + (tx/tz)^2 +
                    // (ty/tz)^2) <= (f/tz)^2 (1 + limx^2 + limy^2), trace(S) = |diag(s) R^T O^T|_F^2 <= |O|_2^2 mf^2
                    const float tzn = __builtin_fminf(__builtin_fabsf(tz_min), __builtin_fabsf(tz_max)) * (1.0f - 1.0e-5f);
                    const float jz = f.focal / tzn;
                    const float mf = A.w * 1.001f;
                    const float trb = jz * jz * (1.0f + f.limx * f.limx + f.limy * f.limy) * f.sigma_vo2 * (mf * mf) * 1.002f + 0.8f;
                    const float hb = 2.0005f * __builtin_fminf(__builtin_sqrtf(2.0f * trb), 5793.0f) + 0.02f;
                    const float slack = 1.0f + 1.0e-4f * (__builtin_fabsf(cxmin) + __builtin_fabsf(cxmax) + __builtin_fabsf(cymin) + __builtin_fabsf(cymax));
                    const float xlo = cxmin - hb - 0.5f - slack, xhi = cxmax + hb - 0.5f + slack;
                    const float ylo = cymin - hb - 0.5f - slack, yhi = cymax + hb - 0.5f + slack;
                    const float wm1 = (float)(f.width - 1), hm1 = (float)(f.height - 1);
                    if (hb < 1.0e9f && xlo < 1.0e9f && xhi > -1.0e9f && ylo < 1.0e9f && yhi > -1.0e9f) {   // (false for NaN / inf)
                        if (xhi < 0.0f || xlo > wm1 || yhi < 0.0f || ylo > hm1) {
                            keep = false;                               // off screen
                        } else {
                            const int tx0 = (int)__builtin_fmaxf(xlo, 0.0f) >> 4, tx1 = (int)__builtin_fminf(xhi, wm1) >> 4;
                            const int ty0 = (int)__builtin_fmaxf(ylo, 0.0f) >> 4, ty1 = (int)__builtin_fminf(yhi, hm1) >> 4;
                            bool behind = false, covered_need = false;
                            float hcl = __builtin_inff();         // the horizon over the tiles the cluster can reach (+inf: none)
                            if (hpyr) {
                                const int r_ = f.cull_dilate;
                                hcl = gsr_pyr_max(hpyr, f.pyr_off, f.tiles_x, max(tx0 - r_, 0), max(ty0 - r_, 0), min(tx1 + r_, f.tiles_x - 1), min(ty1 + r_, f.tiles_y - 1));
                            }
                            if (dact) {
                                // depth-tested frames: does every splat of the cluster lie behind everything the opaque pass left under the
                                // tiles the cluster can reach?  A LOWER bound of its splats' window depths: the smallest z / w of the
                                // corners, less what rounding (here and in K1: eps per clip coordinate) can move a quotient
                                const float zerr = 1.05f * eps * (1.0f + __builtin_fabsf(znmin)) / (wmin - eps) + 2.0e-7f;
                                const float zlo = gsr_fma(znmin - zerr, 0.5f, 0.5f) - 2.0e-7f;
                                // (only where no finite horizon applies: k_preprocess.h says why)
                                behind = !(hcl < 3.0e38f) && zlo > gsr_dpyr_max(dc.pyr, f.pyr_off, f.tiles_x, tx0, ty0, tx1, ty1);
                                // ... or may a COVERED pixel there need one of them (k_preprocess.h: the depth clause of gsr_k1_back)?
                                if (dc.pyrc) {
                                    const int r_ = f.cull_dilate;
                                    covered_need = !(zlo > gsr_dpyr_max(dc.pyrc, f.pyr_off, f.tiles_x, max(tx0 - r_, 0), max(ty0 - r_, 0), min(tx1 + r_, f.tiles_x - 1), min(ty1 + r_, f.tiles_y - 1)));
                                }
                            }
                            if (gsr_owned_rows(ty0, ty1, GsrShard{f.shard_index, f.shard_count, f.shard_rpb}) == 0) {
                                keep = false;                           // none of its tile rows is ours
                            } else if (behind) {
                                keep = false;                           // wholly behind the opaque geometry
                            } else if (hpyr || mode != 0) {
                                // The sort key is the distance^2 of the UN-offset position (k_preprocess.h): bounds from the raw box
                                float d2 = 0.0f, d2far = 0.0f;
#pragma unroll
                                for (int k = 0; k < 3; ++k) {
                                    const float d = __builtin_fmaxf(__builtin_fmaxf(plo[k] - f.cam[k], f.cam[k] - phi[k]), 0.0f);
                                    d2 = gsr_fma(d, d, d2);
                                    const float e = __builtin_fmaxf(__builtin_fabsf(plo[k] - f.cam[k]), __builtin_fabsf(phi[k] - f.cam[k]));
                                    d2far = gsr_fma(e, e, d2far);
                                }
                                d2 *= (1.0f - 1.0e-5f);
                                d2far *= (1.0f + 1.0e-5f);
                                uint32_t kb = __builtin_bit_cast(uint32_t, d2);
                                kb = kb < f.key_min ? f.key_min : (kb > f.key_max ? f.key_max : kb);
                                kb -= f.key_min;
                                kb_near = kb;
                                // front-slab phase 1: the whole cluster lies beyond the slab
                                if (mode == 2 && kb > key_a) keep = false;
                                // front-slab phase 2: every splat of the cluster was drawn by phase 1 (key <= the slab key)
                                if (mode == 3 && d2far < 3.0e38f) {
                                    uint32_t kf = __builtin_bit_cast(uint32_t, d2far);
                                    kf = kf < f.key_min ? f.key_min : (kf > f.key_max ? f.key_max : kf);
                                    if (kf - f.key_min <= key_a) keep = false;
                                }
                                // behind the depth horizon of every tile it can reach (widened by the dilation radius)?
                                if (keep && hpyr && !covered_need && kb > gsr_horizon_key(hcl, f.key_min, f.key_max)) keep = false;
                            }
                        }
                    }
                }
            }
        }
        if (want_hist && keep) {
            const uint32_t b = kb_near >> hist_shift;
            atomicAdd(&s_hist[b < (uint32_t)GSR_SLAB_BINS ? b : (uint32_t)GSR_SLAB_BINS - 1u], 1u);
        }
        // ordered compaction: ballot ranks inside the wave, wave counts through LDS
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_w[wave] = (uint32_t)__builtin_popcountll(bal);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < CC_THREADS / 64; ++w) { const uint32_t c = s_w[w]; before += w < wave ? c : 0u; total += c; }
        if (keep) seg[(size_t)bid * per + kept + before + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull))] = cl;
        kept += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) cnt[bid] = kept;
    if (want_hist) {   // (Morton-ordered clusters: a workgroup's 256 fall into a few dozen bins)
        __syncthreads();
        for (int b = threadIdx.x; b < GSR_SLAB_BINS; b += CC_THREADS) {
            const uint32_t v = s_hist[b];
            if (v) atomicAdd(&slab_hist[b], v);
        }
    }
}

// K1's prologue: the inclusive prefix of the cnt[] of k_cluster_cull, in LDS (ngroups <= CC_MAX_GROUPS), returns the
// number of surviving clusters.  All 256 threads; contains barriers.
__device__ __forceinline__ uint32_t cc_prefix_to_lds(const uint32_t* __restrict__ cnt, uint32_t ngroups, uint32_t* s_inc /*[CC_MAX_GROUPS]*/,
                                                     uint32_t* s_wave /*[4]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < ngroups; base += 256u) {
        const uint32_t g = base + threadIdx.x;
        const uint32_t v = g < ngroups ? cnt[g] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t wb = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const uint32_t t = s_wave[w]; wb += w < wave ? t : 0u; tot += t; }
        if (g < ngroups) s_inc[g] = carry + wb + inc;
        carry += tot;
        __syncthreads();
    }
    return carry;
}
// the rank-th surviving cluster (rank < total): 64-ary search of the inclusive prefix, two LDS reads per lane
__device__ __forceinline__ uint32_t cc_find_cluster(const uint32_t* s_inc, uint32_t ngroups, uint32_t rank,
                                                    const uint32_t* __restrict__ seg, uint32_t per)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t stride = (ngroups + 63u) / 64u;
    // chunk c = groups [c * stride, (c + 1) * stride): the first chunk whose last inclusive prefix exceeds rank
    const uint32_t last = min((lane + 1u) * stride, ngroups) - 1u;
    const bool in1 = lane * stride < ngroups && s_inc[last] > rank;
    const unsigned long long b1 = __ballot(in1);
    const uint32_t c = (uint32_t)__builtin_ctzll(b1 | (1ull << 63));
    uint32_t g = c * stride;
    for (uint32_t base = 0; base < stride; base += 64u) {      // (stride <= 64: one trip)
        const uint32_t gg = c * stride + base + lane;
        const bool in2 = base + lane < stride && gg < ngroups && s_inc[gg] > rank;
        const unsigned long long b2 = __ballot(in2);
        if (b2) { g = c * stride + base + (uint32_t)__builtin_ctzll(b2); break; }
    }
    const uint32_t before = g ? s_inc[g - 1u] : 0u;
    return seg[(size_t)g * per + (rank - before)];
}

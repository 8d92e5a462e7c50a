// gsr_device.h -- shared device-side types and the float32 arithmetic contract
// for the gfx950 GSplat rasterizer.  Compiled with -ffp-contract=off: the only
// fused operations are the explicit __builtin_fmaf() calls below, so every
// threshold decision (w<=0, |q|<=2, alpha<1/255) is reproducible bit for bit.
// DESIGN.md ("Arithmetic contract") is the normative text.
#pragma once
// GSR_KPROF (variant builds only, tools/kprof.py): wall-clock (100 MHz) stamps of workgroup 0 / thread 0 at marked points of the
// small kernels -- where do their microseconds go?
#ifdef GSR_KPROF
#include <hip/hip_runtime.h>
__device__ unsigned long long g_kprof[8][16];
__device__ unsigned int g_kprof_blk[5][2][8192];   // per kernel and workgroup: [0] duration in 10 ns, [1] items
#define KPROF_BLK_BEGIN const unsigned long long kp0_ = wall_clock64();
#define KPROF_BLK_END(k, items) { if (threadIdx.x == 0 && blockIdx.x < 8192) { g_kprof_blk[k][0][blockIdx.x] = (unsigned int)(wall_clock64() - kp0_); g_kprof_blk[k][1][blockIdx.x] = (unsigned int)(items); } }
#define KPROF(k, i) { if (blockIdx.x == 0 && threadIdx.x == 0) g_kprof[k][i] = wall_clock64(); }
#define KPROFB(k, i, blk) { if (blockIdx.x == (blk) && threadIdx.x == 0) g_kprof[k][i] = wall_clock64(); }
#else
#define KPROF(k, i)
#define KPROFB(k, i, blk)
#define KPROF_BLK_BEGIN
#define KPROF_BLK_END(k, items)
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

#define GSR_TILE_PX 16
#define GSR_WAVE 64

// ---- HBM layouts ----------------------------------------------------------
// Geometry, uploaded once (SoA of 16-byte vectors: one coalesced dwordx4 per
// lane per array):
//   geoA[i] = float4 (P.x, P.y, P.z, opacity)
//   geoB[i] = 8 halves (scale.xyz, orient.xyzw, 0)
//   col[c][i], c<6 = 8 halves; half h = 3*j + channel, j = 0 -> Cd, j = 1..15 -> sh_j
//                    (order 0 needs chunk 0, order 1 chunks 0-1, order 2 chunks 0-3,
//                     order 3 chunks 0-5)
//
// Projected record, rewritten every frame (48 B, 3 x float4, AoS so the blend
// kernel's gather touches one or two lines per splat).  Contract v2: the quad-local coordinate of a
// pixel is carried as two affine forms scaled by kappa = sqrt(log2 e),
//     kappa*q0 = d . (a1x, a1y),   kappa*q1 = d . (b1x, b1y),   d = pixel centre - (cx, cy),
// so that exp(-|q|^2) == 2^-(|kappa q|^2) needs no per-fragment 1/s multiply and an exact base-2
// range reduction.
struct __attribute__((aligned(16))) GsrRecord {
    float cx, cy, hx, hy;        // centre (GL window coords), conservative bbox half extents
    float a1x, a1y, b1x, b1y;    // kappa * e / s1, kappa * e_perp / s2  (e = unit major axis, e_perp = (-ey, ex))
    float r, g, b, la;           // colour after SH; log2(opacity) by gsr_log2_opacity() (contract v3: alpha = 2^(la - |kappa q|^2))
};
static_assert(sizeof(GsrRecord) == 48, "record layout");
#define GSR_KAPPA 1.2011224087864498f   // sqrt(log2 e)
#define GSR_QLIM  (2.0f * GSR_KAPPA)    // |q| <= 2 in kappa units

// Frame constants, computed once per frame on the host in float32 (same
// operation order as the oracle) and passed by value.
struct GsrFrame {
    float ov[12];      // rows 0..2 of glH_ObjViewMatrix: ov[r*4+c]
    float pr[16];      // glH_ProjectMatrix rows: pr[r*4+c]
    float vw[12];      // rows 0..2 of glH_ViewMatrix
    float ob[9];       // mat3(glH_ObjectMatrix) rows: ob[r*3+c]
    float io[9];       // mat3(glH_InvObjectMatrix) rows
    float cam[3];
    float origin[3];
    float limx, limy;  // 1.3*tanFovX, 1.3*tanFovY
    float sigma_vo2;   // upper bound of (largest singular value of mat3(view) x that of mat3(object))^2: K1's early ownership test
    float focal;       // (W*P00)*0.5
    float W, H;
    int32_t width, height;
    int32_t sh_order;  // already gated by SH presence
    int32_t tiles_x;   // ceil(width/16)
    int32_t tiles_y;   // ceil(height/16) (whole image)
    int32_t shard_index, shard_count;  // which tile rows are ours: see GsrShard
    int32_t shard_rpb;                 // 0 = interleaved rows (r % count == index); > 0 = contiguous bands of rpb tile rows
    int32_t local_tiles_y;             // rows owned by this shard
    int32_t super;                     // super-tile edge in tiles (power of two)
    int32_t super_shift;               // log2(super)
    int32_t stiles_x, stiles_y;        // super-tile grid over the whole image
    int32_t flags;                     // GSR_FLAG_* (A/B switches; never change pixels)
    uint32_t key_min, key_max;         // sort keys are stored as clamp(bits, key_min, key_max) - key_min
    int32_t pyr_off[6];                // depth-horizon pyramid: first cell of level l (k_cluster.h; GSR_PYR_LEVELS)
    int32_t cull_dilate;               // tiles by which a rect is widened before it is compared with the horizons (on top of the
                                       // dilation built into the pyramid)
    int32_t rect_shift;                // tile rects are packed in units of (1 << rect_shift) tiles: 0 up to 256 tiles a side, 1 up to 512, 2 up to 1024
    int32_t phase;                     // front-slab frames (gsr_api.hip): 0 = the whole frame; 1 = only the splats with key <= the slab key
                                       // (device word, picked by k_slab_pick); 2 = only the ones beyond it, culled against the tiles
                                       // that phase 1 left opaque
    // Depth-tested frames of clouds below 2^23 splats (the reference's own limit): the nine spare bits of a list entry's splat index carry a
    // coarse WINDOW DEPTH (gsr_zq: monotone, a lower bound), so that a tile can drop the entries that lie behind everything the opaque pass
    // left under its live pixels while it SCANS its list -- before they cost a queue slot, a 48-byte gather and a quadrant test per wave
    // (k_bin_place writes the bits, k_blend reads them; everybody else masks them off).  idx_mask = 0xffffffff: no such bits.
    uint32_t idx_mask;
    float zq0, zqs;                    // code = clamp(floor((zwin - zq0) * zqs), 0, 511)
};
#define GSR_ZQ_SHIFT 23
#define GSR_ZQ_MAX 511.0f
#define GSR_SLAB_BINS 1024             // k_cluster_cull's histogram of the surviving clusters' nearest keys (k_slab_pick reads it)
#define GSR_FLAG_NO_ALPHA_RADIUS 1   // bbox from the full +-2 quad instead of the alpha>=1/255 support
#define GSR_FLAG_NO_SAT          2   // quadrant masks from the bbox only
#define GSR_FLAG_FULL_KEYS       4   // sort all 32 key bits (no key-range reduction, 8-bit digits)
#define GSR_FLAG_LAZY_NO_PREFIX  8   // lazy colour: colour nothing ahead of time, so that every tile takes the on-demand fallback
#define GSR_FLAG_CULL_ROUNDS    16   // k_cluster_cull: three rounds of clusters per workgroup (what clouds beyond 33 M splats do)
#define GSR_FLAG_NO_DEPTH_CLASS 32   // depth-tested frames: no per-quadrant classification in k_blend (every record staged, every fragment compared)
#define GSR_FLAG_NO_ZCODES      64   // depth-tested frames: no coarse window depth in the list entries' spare index bits (GsrFrame.idx_mask)

// small-frame depth sort (k_sort.h): bucket regions
#ifndef BK_BUCKETS
#define BK_BUCKETS 1024                      // (measured on C4: 512 buckets x 4096-key chunks 45 us, 1024 x 2048 31 us, 2048 x 1024 30 us)
#endif
#ifndef BK_CAP
#define BK_CAP 8192                          // slots of a bucket's region
#endif
#define BK_STRIDE 64                         // uint32 between two bucket counters (256 bytes: atomics on one line serialise)

// ---- scalar helpers ---------------------------------------------------------
__device__ __forceinline__ float gsr_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// nine-bit code of a window depth, monotone non-decreasing (so code(z) > code(limit) implies z > limit); NaN -> 0, +inf -> 511
__device__ __forceinline__ uint32_t gsr_zq(float zw, float z0, float zs)
{
    const float t = __builtin_fminf(__builtin_fmaxf((zw - z0) * zs, 0.0f), GSR_ZQ_MAX);
    return (uint32_t)t;
}

__device__ __forceinline__ float gsr_h2f(uint32_t bits16)
{
    _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)bits16);
    return (float)h;  // v_cvt_f32_f16: exact
}

// The contract's 2^x for x in [-2^22, 0]: identical operation sequence to the oracle's gso_exp2f
// (round to the nearest-even integer with the 1.5*2^23 trick, EXACT remainder r = x - k, degree-5
// polynomial for 2^r on [-0.5, 0.5], exponent add).  <= 2.8 ulp, exp2(0) == 1, never above 1.
__device__ __forceinline__ float gsr_exp2n(float x)
{
    const float magic = 12582912.0f;
    const float s = x + magic;
    const float kf = s - magic;
    const float r = x - kf;
    float p = 1.3292919611558318e-3f;
    p = gsr_fma(p, r, 9.671509265899658e-3f);
    p = gsr_fma(p, r, 5.550636723637581e-2f);
    p = gsr_fma(p, r, 2.4022242426872253e-1f);
    p = gsr_fma(p, r, 6.931470632553101e-1f);
    const float y = gsr_fma(p, r, 1.0f);
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, y) + (__builtin_bit_cast(uint32_t, s) << 23));
}

// two-wide float: the blend kernel evaluates two records per lane per iteration so that the
// compiler can emit packed-FP32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32); each
// component is still an IEEE fma.
typedef float gsr_v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ gsr_v2f gsr_fma2(gsr_v2f a, gsr_v2f b, gsr_v2f c) { return __builtin_elementwise_fma(a, b, c); }

// gsr_exp2n on the two lanes of a pair: the same values per component (the integer k sits in the low mantissa
// bits of s; shifting them into the exponent field discards the rest).
__device__ __forceinline__ gsr_v2f gsr_exp2n2(gsr_v2f x)
{
    const gsr_v2f magic = (gsr_v2f)(12582912.0f);
    const gsr_v2f s = x + magic;
    const gsr_v2f kf = s - magic;
    const gsr_v2f r = x - kf;
    gsr_v2f p = (gsr_v2f)(1.3292919611558318e-3f);
    p = gsr_fma2(p, r, (gsr_v2f)(9.671509265899658e-3f));
    p = gsr_fma2(p, r, (gsr_v2f)(5.550636723637581e-2f));
    p = gsr_fma2(p, r, (gsr_v2f)(2.4022242426872253e-1f));
    p = gsr_fma2(p, r, (gsr_v2f)(6.931470632553101e-1f));
    const gsr_v2f y = gsr_fma2(p, r, (gsr_v2f)(1.0f));
    // NB: __builtin_bit_cast applied directly to a vector ELEMENT (y.y) reads element 0 with this
    // compiler (ROCm 7.2 clang) -- go through scalar temporaries.
    const float y0 = y.x, y1 = y.y, s0 = s.x, s1 = s.y;
    gsr_v2f o;
    o.x = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, y0) + (__builtin_bit_cast(uint32_t, s0) << 23));
    o.y = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, y1) + (__builtin_bit_cast(uint32_t, s1) << 23));
    return o;
}

// Largest |q| at which alpha = exp(-|q|^2)*opacity can still reach 1/255 (capped at the quad's
// half width 2).  Conservative: fast log (abs error < 1e-5 here) plus margins, so it only ever
// removes pixels the fragment test would discard anyway.  Not part of the parity contract.
// On the native v_log_f32 / v_sqrt_f32 (1 ulp each; the IEEE versions are ~45 instructions): the radius only feeds conservative
// culling (K1's bbox, the blend kernel's quadrant test), and the 1e-3 margins dwarf the difference.
__device__ __forceinline__ float gsr_support_radius_fast(float opacity)
{
    const float L = __builtin_amdgcn_logf(255.0f * opacity) * 0.693147181f;
    return __builtin_fminf(2.0f, __builtin_amdgcn_sqrtf(__builtin_fmaxf(L, 0.0f) + 1.0e-4f) + 1.0e-3f);
}
// the same radius from la = log2(opacity): ln(255 opacity) = (la + log2 255) ln 2  (+1e-5: la carries ~1e-7 of its own error)
__device__ __forceinline__ float gsr_support_radius_from_la(float la)
{
    const float L = (la + 7.99435343685886f) * 0.693147181f + 1.0e-5f;
    return __builtin_fminf(2.0f, __builtin_amdgcn_sqrtf(__builtin_fmaxf(L, 0.0f) + 1.0e-4f) + 1.0e-3f);
}

// Contract v3: alpha lives in the log domain.  la = log2(opacity), formed once per splat with exactly the operations of the
// oracle's gso_log2_opacity() (oracle/gsplat_oracle.c): -inf unless opacity >= 1/255 (NaN too), +inf for +inf, else
// e + 2/ln2 * atanh((m-1)/(m+1)), opacity = m 2^e, m in (sqrt(1/2), sqrt(2)], atanh by its series up to s^9.  The fragment's
// alpha is 2^(la - pw) -- one subtraction and one v_exp_f32 in k_blend -- and it is discarded iff la - pw < -log2(255): the
// decision is taken on the argument, which oracle and kernel compute bit-identically.
#define GSR_LOG2_255 7.99435343685886f
__device__ __forceinline__ float gsr_log2_opacity(float opacity)
{
    if (!(opacity >= (1.0f / 255.0f))) return -__builtin_inff();
    if (opacity > 3.4028234663852886e38f) return __builtin_inff();
    uint32_t bits = __builtin_bit_cast(uint32_t, opacity);
    int32_t e = (int32_t)(bits >> 23) - 127;
    float m = __builtin_bit_cast(float, (bits & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float s = (m - 1.0f) / (m + 1.0f);      // IEEE division (-ffp-contract=off, no fast-math)
    const float z = s * s;
    float p = 1.0f / 9.0f;
    p = __builtin_fmaf(p, z, 1.0f / 7.0f);
    p = __builtin_fmaf(p, z, 1.0f / 5.0f);
    p = __builtin_fmaf(p, z, 1.0f / 3.0f);
    p = __builtin_fmaf(p, z, 1.0f);
    const float l = s * p;
    return __builtin_fmaf(l, 2.885390043f, (float)e);
}

// A depth horizon (distance^2 from the camera; +inf or NaN = none) in a frame's sort-key domain: K1 and the binning kernels
// compare keys with this, so they agree exactly on what lies beyond it.
__host__ __device__ __forceinline__ uint32_t gsr_horizon_key(float h, uint32_t key_min, uint32_t key_max)
{
    if (!(h < 3.0e38f)) return 0xffffffffu;
    uint32_t b = __builtin_bit_cast(uint32_t, h > 0.0f ? h : 0.0f);
    b = b < key_min ? key_min : (b > key_max ? key_max : b);
    return b - key_min;
}

// rect packing: four 8-bit coordinates in RECT UNITS of (1 << g) tiles, g = GsrFrame.rect_shift: tile coords < 256 << g.
// A frame of more than 4096 pixels a side (g = 1) carries rects rounded outwards to pairs of tiles: everything that reads
// a rect -- ownership, horizons, super-tile lists, the blend kernel's tile filter -- errs towards "touches", and the exact
// test against the pixels happens in k_blend as always.
__device__ __forceinline__ uint32_t gsr_pack_rect(int x0, int y0, int x1, int y1)
{
    return (uint32_t)x0 | ((uint32_t)y0 << 8) | ((uint32_t)x1 << 16) | ((uint32_t)y1 << 24);
}
#define GSR_RECT_EMPTY 0x00000001u  // x0=1 > x1=0

// Tile-row ownership of a rank.  Two layouts:
//   interleaved (rpb == 0): row r belongs to rank r % count -- balances any scene, but almost every splat reaches a row of
//                           almost every rank once count is small against the splat's height in rows;
//   bands       (rpb  > 0): rank g owns rows [g*rpb, (g+1)*rpb) -- a rank keeps ~1/count of the splats (plus a boundary
//                           strip), and a splat's centre row alone usually tells whether it is ours (early-out in K1).
struct GsrShard {
    int32_t index, count, rpb;
    int32_t g;                         // GsrFrame.rect_shift (rects -> tile rows), 0 where no rect is involved
};
__host__ __device__ __forceinline__ bool gsr_shard_owns(const GsrShard& sh, int r)
{
    return sh.rpb > 0 ? (r / sh.rpb == sh.index) : (r % sh.count == sh.index);
}
// global tile row of the rank's local row l
__host__ __device__ __forceinline__ int gsr_shard_global_row(const GsrShard& sh, int l)
{
    return sh.rpb > 0 ? sh.index * sh.rpb + l : l * sh.count + sh.index;
}
// number of tile rows in [y0,y1] owned by the shard
__host__ __device__ __forceinline__ int gsr_owned_rows(int y0, int y1, const GsrShard& sh)
{
    if (sh.rpb > 0) {
        const int lo = sh.index * sh.rpb, hi = lo + sh.rpb - 1;
        const int a = y0 > lo ? y0 : lo, b = y1 < hi ? y1 : hi;
        return b >= a ? b - a + 1 : 0;
    }
    // first owned row >= y0
    const int first = y0 + ((sh.index - y0 % sh.count) + sh.count) % sh.count;
    if (first > y1) return 0;
    return (y1 - first) / sh.count + 1;
}

// tile rows [lo, hi] of the rect rows [y0, y1]
__device__ __forceinline__ int gsr_owned_rect_rows(int y0, int y1, const GsrShard& sh)
{
    return gsr_owned_rows(y0 << sh.g, ((y1 + 1) << sh.g) - 1, sh);
}
// 0 iff the shard owns none of the rect's rows (otherwise an upper bound of its tiles there)
__device__ __forceinline__ int gsr_rect_tiles(uint32_t rect, const GsrShard& sh)
{
    int x0 = rect & 255, y0 = (rect >> 8) & 255, x1 = (rect >> 16) & 255, y1 = rect >> 24;
    if (x1 < x0 || y1 < y0) return 0;
    return ((x1 - x0 + 1) << sh.g) * gsr_owned_rect_rows(y0, y1, sh);
}

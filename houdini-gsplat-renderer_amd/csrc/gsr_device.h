// gsr_device.h -- shared device-side types and the float32 arithmetic contract
// for the gfx950 GSplat rasterizer.  Compiled with -ffp-contract=off: the only
// fused operations are the explicit __builtin_fmaf() calls below, so every
// threshold decision (w<=0, |q|<=2, alpha<1/255) is reproducible bit for bit.
// DESIGN.md ("Arithmetic contract") is the normative text.
#pragma once

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
// kernel's gather touches one or two lines per splat):
struct __attribute__((aligned(16))) GsrRecord {
    float cx, cy, ex, ey;        // centre (GL window coords), unit major axis
    float is1, is2, hx, hy;      // 1/s1, 1/s2, conservative bbox half extents
    float r, g, b, opacity;      // colour after SH, opacity
};
static_assert(sizeof(GsrRecord) == 48, "record layout");

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
    float focal;       // (W*P00)*0.5
    float W, H;
    int32_t width, height;
    int32_t sh_order;  // already gated by SH presence
    int32_t tiles_x;   // ceil(width/16)
    int32_t tiles_y;   // ceil(height/16) (whole image)
    int32_t shard_index, shard_count;  // tile row r is ours iff r % count == index
    int32_t local_tiles_y;             // rows owned by this shard
    int32_t super;                     // super-tile edge in tiles (power of two)
    int32_t super_shift;               // log2(super)
    int32_t stiles_x, stiles_y;        // super-tile grid over the whole image
    int32_t flags;                     // GSR_FLAG_* (A/B switches; never change pixels)
    uint32_t key_min, key_max;         // sort keys are stored as clamp(bits, key_min, key_max) - key_min
};
#define GSR_FLAG_NO_ALPHA_RADIUS 1   // bbox from the full +-2 quad instead of the alpha>=1/255 support
#define GSR_FLAG_NO_SAT          2   // quadrant masks from the bbox only
#define GSR_FLAG_FULL_KEYS       4   // sort all 32 key bits (no key-range reduction, 8-bit digits)

// ---- scalar helpers ---------------------------------------------------------
__device__ __forceinline__ float gsr_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float gsr_h2f(uint32_t bits16)
{
    _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)bits16);
    return (float)h;  // v_cvt_f32_f16: exact
}

// The contract's exp() for x in [-80, 0]: identical operation sequence to the
// oracle's gso_expf (range reduction by ln2 hi/lo, degree-5 polynomial, exponent add).
__device__ __forceinline__ float gsr_expf(float x)
{
    float kf = __builtin_rintf(x * 1.44269504088896341f);
    float r = gsr_fma(kf, -0.693359375f, x);
    r = gsr_fma(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = gsr_fma(p, r, 1.3981999507e-3f);
    p = gsr_fma(p, r, 8.3334519073e-3f);
    p = gsr_fma(p, r, 4.1665795894e-2f);
    p = gsr_fma(p, r, 1.6666665459e-1f);
    p = gsr_fma(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = gsr_fma(p, r2, r) + 1.0f;
    int32_t k = (int32_t)kf;
    uint32_t bits = __builtin_bit_cast(uint32_t, y) + ((uint32_t)k << 23);
    return __builtin_bit_cast(float, bits);
}

// two-wide float: the blend kernel evaluates two records per lane per iteration so that the
// compiler can emit packed-FP32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 -- the
// only way to reach gfx950's 157 TFLOP/s vector peak); each component is still an IEEE fma.
typedef float gsr_v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ gsr_v2f gsr_fma2(gsr_v2f a, gsr_v2f b, gsr_v2f c) { return __builtin_elementwise_fma(a, b, c); }

// gsr_expf on two lanes of a pair: the same values per component.  rint() and the float->int conversion are
// half-rate, unpackable instructions on gfx950 (tools/ubench_valu.hip), so the rounding is done the classic
// way: s = t + 1.5*2^23 rounds t to the nearest-even integer in s's low mantissa bits (|t| < 2^22; here
// t in [-116, 0]), s - 1.5*2^23 is that integer as a float (== rintf(t)), and the integer itself is the low
// bits of s -- shifting them into the exponent field discards the rest.  Bit-identical to gsr_expf.
__device__ __forceinline__ gsr_v2f gsr_expf2(gsr_v2f x)
{
    const gsr_v2f t = x * 1.44269504088896341f;
    const gsr_v2f magic = (gsr_v2f)(12582912.0f);
    const gsr_v2f s = t + magic;
    const gsr_v2f kf = s - magic;
    gsr_v2f r = gsr_fma2(kf, (gsr_v2f)(-0.693359375f), x);
    r = gsr_fma2(kf, (gsr_v2f)(2.12194440e-4f), r);
    gsr_v2f p = (gsr_v2f)(1.9875691500e-4f);
    p = gsr_fma2(p, r, (gsr_v2f)(1.3981999507e-3f));
    p = gsr_fma2(p, r, (gsr_v2f)(8.3334519073e-3f));
    p = gsr_fma2(p, r, (gsr_v2f)(4.1665795894e-2f));
    p = gsr_fma2(p, r, (gsr_v2f)(1.6666665459e-1f));
    p = gsr_fma2(p, r, (gsr_v2f)(5.0000001201e-1f));
    const gsr_v2f r2 = r * r;
    const gsr_v2f y = gsr_fma2(p, r2, r) + 1.0f;
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
__device__ __forceinline__ float gsr_support_radius(float opacity)
{
    const float L = __logf(255.0f * opacity);
    return __builtin_fminf(2.0f, __builtin_sqrtf(__builtin_fmaxf(L, 0.0f) + 1.0e-4f) + 1.0e-3f);
}

// rect packing: tile coords < 256 (GSR_MAX_DIM 4096 / 16)
__device__ __forceinline__ uint32_t gsr_pack_rect(int x0, int y0, int x1, int y1)
{
    return (uint32_t)x0 | ((uint32_t)y0 << 8) | ((uint32_t)x1 << 16) | ((uint32_t)y1 << 24);
}
#define GSR_RECT_EMPTY 0x00000001u  // x0=1 > x1=0

// number of tile rows in [y0,y1] owned by shard (index,count): rows r with r%count==index
__device__ __forceinline__ int gsr_owned_rows(int y0, int y1, int index, int count)
{
    // first owned row >= y0
    int first = y0 + ((index - y0 % count) + count) % count;
    if (first > y1) return 0;
    return (y1 - first) / count + 1;
}

__device__ __forceinline__ int gsr_rect_tiles(uint32_t rect, int index, int count)
{
    int x0 = rect & 255, y0 = (rect >> 8) & 255, x1 = (rect >> 16) & 255, y1 = rect >> 24;
    if (x1 < x0 || y1 < y0) return 0;
    return (x1 - x0 + 1) * gsr_owned_rows(y0, y1, index, count);
}

// number of SUPER-tiles (2^shift x 2^shift tiles) the rect reaches through at least one owned tile row
__device__ __forceinline__ int gsr_rect_supers(uint32_t rect, int shift, int index, int count)
{
    const int x0 = rect & 255, y0 = (rect >> 8) & 255, x1 = (rect >> 16) & 255, y1 = rect >> 24;
    if (x1 < x0 || y1 < y0) return 0;
    const int w = (x1 >> shift) - (x0 >> shift) + 1;
    const int sy0 = y0 >> shift, sy1 = y1 >> shift;
    if (count == 1) return w * (sy1 - sy0 + 1);
    int rows = 0;
    for (int sy = sy0; sy <= sy1; ++sy) {
        const int lo = max(y0, sy << shift), hi = min(y1, ((sy + 1) << shift) - 1);
        rows += gsr_owned_rows(lo, hi, index, count) > 0 ? 1 : 0;
    }
    return w * rows;
}

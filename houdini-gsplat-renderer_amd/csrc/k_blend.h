// k_blend.h -- K6: per-tile front-to-back compositing.
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
// Data flow: the tile's sorted pair list is consumed in chunks of 256; each
// thread gathers one 48-B record (3 x dwordx4) into registers one chunk ahead,
// drops it into a double-buffered LDS stage together with a 4-bit
// quadrant-overlap mask, and every wave turns those masks into a 64-bit ballot
// so that it only ever touches records whose bbox reaches its quadrant.
// Early-out: a wave stops when all its pixels have 1-A < 2^-14 (error bound
// 2^-14 * max colour, well inside the 1e-3 budget); the workgroup stops
// fetching when all four waves have stopped.
#pragma once
#include "gsr_device.h"

#define BL_CHUNK 256
#define GSR_T_MIN 6.103515625e-05f  // 2^-14

struct GsrBlendArgs {
    int32_t width, height;      // full image
    int32_t tiles_x;            // tiles per row
    int32_t local_tiles;        // tiles_x * local_tiles_y
    int32_t shard_index, shard_count;
    int32_t band_rows;          // pixel rows of the output (band) image
    int32_t swizzle;            // XCD-aware tile mapping
    int32_t swz_chunk;          // tiles per XCD when swizzled
};

// blockIdx -> tile.  Workgroup b runs on XCD b%8 (observed dispatch order,
// MI355X guide): give each XCD a contiguous run of tiles so that neighbouring
// tiles -- which gather the same records -- share one L2.
__device__ __forceinline__ int gsr_tile_of_block(const GsrBlendArgs& a)
{
    const int b = blockIdx.x;
    if (!a.swizzle) return b;
    return (b & 7) * a.swz_chunk + (b >> 3);
}

__global__ void __launch_bounds__(256)
k_blend(GsrBlendArgs a, const uint32_t* __restrict__ pvals, const int32_t* __restrict__ tstart,
        const int32_t* __restrict__ tend, const GsrRecord* __restrict__ recs, float4* __restrict__ out,
        uint32_t* __restrict__ tile_loaded)
{
    __shared__ float4 s0[2][BL_CHUNK];   // cx, cy, ex, ey
    __shared__ float4 s1[2][BL_CHUNK];   // is1, is2, (unused hx, hy)
    __shared__ float4 s2[2][BL_CHUNK];   // r, g, b, opacity
    __shared__ uint32_t smask[2][BL_CHUNK];
    __shared__ uint32_t sdone[2][4];

    const int tile = gsr_tile_of_block(a);
    if (tile >= a.local_tiles) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tile % a.tiles_x, lty = tile / a.tiles_x;
    const int gty = lty * a.shard_count + a.shard_index;
    const int qx = tx * GSR_TILE_PX + (wave & 1) * 8, qy = gty * GSR_TILE_PX + (wave >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const bool pix_ok = (px < a.width) && (py < a.height);
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    // tile bounds in pixel-centre coordinates, for the quadrant masks
    const float tcx0 = (float)(tx * GSR_TILE_PX) + 0.5f, tcy0 = (float)(gty * GSR_TILE_PX) + 0.5f;

    float C0 = 0.0f, C1 = 0.0f, C2 = 0.0f, A = 0.0f;
    const int s = tstart[tile];
    const int n = tend[tile] - s;
    bool wave_done = false;
    int loaded = 0;

    float4 r0, r1, r2;
    bool have = tid < n;
    if (have) {
        const float4* p = reinterpret_cast<const float4*>(recs + pvals[s + tid]);
        r0 = p[0]; r1 = p[1]; r2 = p[2];
    }
    loaded = n < BL_CHUNK ? n : BL_CHUNK;

    for (int c = 0; c * BL_CHUNK < n; ++c) {
        const int buf = c & 1;
        if (have) {
            s0[buf][tid] = r0; s1[buf][tid] = r1; s2[buf][tid] = r2;
            // which 8x8 quadrants can the splat's bbox touch?
            const float bx0 = r0.x - r1.z, bx1 = r0.x + r1.z, by0 = r0.y - r1.w, by1 = r0.y + r1.w;
            const bool xl = bx0 <= tcx0 + 7.0f, xr = bx1 >= tcx0 + 8.0f;
            const bool yb = by0 <= tcy0 + 7.0f, yt = by1 >= tcy0 + 8.0f;
            smask[buf][tid] = (uint32_t)(xl && yb) | ((uint32_t)(xr && yb) << 1) | ((uint32_t)(xl && yt) << 2) |
                              ((uint32_t)(xr && yt) << 3);
        }
        if (lane == 0) sdone[buf][wave] = wave_done ? 1u : 0u;
        __syncthreads();
        const bool block_done = (sdone[buf][0] & sdone[buf][1] & sdone[buf][2] & sdone[buf][3]) != 0u;
        if (block_done) break;
        const int cn = (n - c * BL_CHUNK < BL_CHUNK) ? (n - c * BL_CHUNK) : BL_CHUNK;

        // prefetch the next chunk while this one is composited
        const int nxt = (c + 1) * BL_CHUNK + tid;
        have = nxt < n;
        if (have) {
            const float4* p = reinterpret_cast<const float4*>(recs + pvals[s + nxt]);
            r0 = p[0]; r1 = p[1]; r2 = p[2];
        }
        if ((c + 1) * BL_CHUNK < n) {
            const int more = n - (c + 1) * BL_CHUNK;
            loaded += more < BL_CHUNK ? more : BL_CHUNK;
        }

        if (!wave_done) {
            for (int g = 0; g * 64 < cn; ++g) {
                const int j0 = g * 64;
                const bool mine = (j0 + lane < cn) && ((smask[buf][j0 + lane] >> wave) & 1u);
                unsigned long long acc = __ballot(mine);
                while (acc) {
                    const int j = j0 + __builtin_ctzll(acc);
                    acc &= acc - 1;
                    const float4 g0 = s0[buf][j];
                    const float4 g1 = s1[buf][j];
                    const float4 g2 = s2[buf][j];
                    const float dx = fx - g0.x, dy = fy - g0.y;
                    const float u = gsr_fma(dx, g0.z, dy * g0.w);
                    const float v = gsr_fma(dy, g0.z, -(dx * g0.w));
                    const float q0 = u * g1.x, q1 = v * g1.y;
                    const bool inside = (__builtin_fabsf(q0) <= 2.0f) && (__builtin_fabsf(q1) <= 2.0f);
                    const float power = -gsr_fma(q0, q0, q1 * q1);
                    // outside the quad power can be very negative: keep the exp argument in range
                    float alpha = gsr_expf(__builtin_fmaxf(power, -80.0f)) * g2.w;
                    alpha = __builtin_fminf(__builtin_fmaxf(alpha, 0.0f), 1.0f);
                    if (inside && alpha >= (1.0f / 255.0f)) {
                        const float t = 1.0f - A;
                        C0 = gsr_fma(t, g2.x * alpha, C0);
                        C1 = gsr_fma(t, g2.y * alpha, C1);
                        C2 = gsr_fma(t, g2.z * alpha, C2);
                        A = gsr_fma(t, alpha, A);
                    }
                }
                if (__all(!pix_ok || (1.0f - A) < GSR_T_MIN)) { wave_done = true; break; }
            }
        }
    }
    if (pix_ok) {
        const int brow = lty * GSR_TILE_PX + (wave >> 1) * 8 + (lane >> 3);
        out[(size_t)brow * a.width + px] = make_float4(C0, C1, C2, A);
    }
    if (tid == 0) tile_loaded[tile] = (uint32_t)loaded;  // pairs fetched by this tile (D_eff bookkeeping)
}

// D_eff bookkeeping: one workgroup sums the per-tile fetch counts into counters[1] (this
// frame) and counters[2] (running total for bench.py's roofline) -- two atomics per frame
// instead of two per tile.
__global__ void __launch_bounds__(256)
k_sum_loaded(const uint32_t* __restrict__ tile_loaded, int n_tiles, unsigned long long* __restrict__ counters)
{
    __shared__ unsigned long long s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    unsigned long long local = 0;
    for (int i = threadIdx.x; i < n_tiles; i += 256) local += tile_loaded[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) local += __shfl_down(local, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_sum, local);
    __syncthreads();
    if (threadIdx.x == 0) {
        counters[1] = s_sum;
        atomicAdd(&counters[2], s_sum);
    }
}

// k_binning.h -- splat -> tile duplication in depth order, and per-tile ranges.
//
// The GL rasteriser did this implicitly for the reference (one instanced quad
// per splat, /root/reference/gsplat_plugin/src/GSplatRenderer.C:647); a tiled
// compute rasterizer has to materialise (tile, splat) pairs.  Pairs are emitted
// in DEPTH-RANK order, so a stable sort on the tile id alone yields per-tile
// front-to-back lists.  Roofline: HBM (8 B written per pair; 12 B read per splat).
#pragma once
#include "gsr_device.h"

// cnt[r] = number of owned tiles of the splat at depth rank r
__global__ void __launch_bounds__(256)
k_tile_counts(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ rect, uint32_t n,
              int shard_index, int shard_count, uint32_t* __restrict__ cnt)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n) return;
    cnt[r] = (uint32_t)gsr_rect_tiles(rect[perm[r]], shard_index, shard_count);
}

// one lane per depth rank; small rects are written by their lane, rects with
// more than 32 tiles are written cooperatively by the whole wave (giant splats:
// the axis cap is 4096 px)
__global__ void __launch_bounds__(256)
k_emit_pairs(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ rect,
             const uint32_t* __restrict__ poff, uint32_t n, int shard_index, int shard_count, int tiles_x,
             uint32_t* __restrict__ pkeys, uint32_t* __restrict__ pvals)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t idx = 0, rc = GSR_RECT_EMPTY, o = 0;
    int cnt = 0;
    if (r < n) {
        idx = perm[r];
        rc = rect[idx];
        cnt = gsr_rect_tiles(rc, shard_index, shard_count);
        o = poff[r];
    }
    const bool big = cnt > 32;
    if (cnt > 0 && !big) {
        const int x0 = rc & 255, y0 = (rc >> 8) & 255, x1 = (rc >> 16) & 255, y1 = rc >> 24;
        int ty = y0 + ((shard_index - y0 % shard_count) + shard_count) % shard_count;
        for (; ty <= y1; ty += shard_count) {
            const uint32_t rowkey = (uint32_t)(ty / shard_count) * (uint32_t)tiles_x;
            for (int tx = x0; tx <= x1; ++tx) {
                pkeys[o] = rowkey + (uint32_t)tx;
                pvals[o] = idx;
                ++o;
            }
        }
    }
    unsigned long long m = __ballot(big);
    while (m) {
        const int src = __builtin_ctzll(m);
        m &= m - 1;
        const uint32_t rc_s = __shfl(rc, src, 64);
        const uint32_t o_s = __shfl(o, src, 64);
        const uint32_t idx_s = __shfl(idx, src, 64);
        const int x0 = rc_s & 255, y0 = (rc_s >> 8) & 255, x1 = (rc_s >> 16) & 255, y1 = rc_s >> 24;
        const int w = x1 - x0 + 1;
        const int first = y0 + ((shard_index - y0 % shard_count) + shard_count) % shard_count;
        const int rows = (first > y1) ? 0 : (y1 - first) / shard_count + 1;
        const int total = w * rows;
        for (int t = lane; t < total; t += 64) {
            const int ry = t / w, tx = x0 + t % w;
            const int ty = first + ry * shard_count;
            pkeys[o_s + t] = (uint32_t)(ty / shard_count) * (uint32_t)tiles_x + (uint32_t)tx;
            pvals[o_s + t] = idx_s;
        }
    }
}

// boundaries of equal-key runs in the tile-sorted pair list
__global__ void __launch_bounds__(256)
k_tile_ranges(const uint32_t* __restrict__ keys, uint32_t n, int32_t* __restrict__ tstart, int32_t* __restrict__ tend)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    if (i == 0 || keys[i - 1] != k) tstart[k] = (int32_t)i;
    if (i == n - 1 || keys[i + 1] != k) tend[k] = (int32_t)(i + 1);
}

// root side of the multi-GPU path: de-interleave gathered band images.
// gathered = count bands of band_rows x width pixels; band g holds tile rows
// g, g+count, ... stacked bottom-up.
__global__ void __launch_bounds__(256)
k_stitch_bands(const float4* __restrict__ gathered, int count, int band_rows, int width, int height,
               float4* __restrict__ out)
{
    const size_t p = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (p >= (size_t)width * height) return;
    const int x = (int)(p % width), y = (int)(p / width);
    const int trow = y >> 4, g = trow % count, lrow = trow / count;
    const int by = lrow * GSR_TILE_PX + (y & 15);
    out[p] = gathered[((size_t)g * band_rows + by) * width + x];
}

// k_binning.h -- coarse binning: splat -> SUPER-tile pairs in depth order, and
// per-super-tile ranges.
//
// The GL rasteriser did all binning implicitly for the reference (one instanced
// quad per splat, /root/reference/gsplat_plugin/src/GSplatRenderer.C:647).  Here
// only a COARSE (super-tile = SxS tiles, <= 256 of them) list is materialised:
// pairs are emitted in depth-rank order, ONE stable 8-bit radix pass on the
// super-tile id turns them into front-to-back lists, and the blend kernel
// filters each list down to its own 16x16 tile on the fly (wave ballots), so it
// stops reading the moment the tile is opaque.  Materialising per-tile lists
// instead cost 75 M pairs x 2 radix passes on the 6 M-splat scene, 93 % of which
// were never consumed (profiles/r1_baseline_v1).
// Roofline: HBM (12 B written per pair; 12 B read per splat, all coalesced).
#pragma once
#include "gsr_device.h"

// The depth sort carries a uint2 payload (splat index, packed tile rect), so everything
// below reads its inputs coalesced in depth-rank order -- no gathers.

// cnt[r] = number of owned super-tiles of the splat at depth rank r
__global__ void __launch_bounds__(256)
k_super_counts(const uint2* __restrict__ sorted, uint32_t n_max, const uint32_t* __restrict__ n_dev, int shift,
               int shard_index, int shard_count, uint32_t* __restrict__ cnt)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n_max) return;
    // n_dev = number of splats that survived the compacting first sort pass; the scan runs over n_max
    cnt[r] = (r < *n_dev) ? (uint32_t)gsr_rect_supers(sorted[r].y, shift, shard_index, shard_count) : 0u;
}

// one lane per depth rank writes its (super-tile id ; splat index, rect) pairs at poff[r]
__global__ void __launch_bounds__(256)
k_emit_pairs(const uint2* __restrict__ sorted, const uint32_t* __restrict__ poff, const uint32_t* __restrict__ n_dev,
             int shift, int shard_index, int shard_count, int stiles_x, uint32_t* __restrict__ pkeys,
             uint2* __restrict__ pvals)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= *n_dev) return;
    const uint2 v = sorted[r];
    const uint32_t rc = v.y;
    const int x0 = rc & 255, y0 = (rc >> 8) & 255, x1 = (rc >> 16) & 255, y1 = rc >> 24;
    if (x1 < x0 || y1 < y0) return;
    uint32_t o = poff[r];
    const int sx0 = x0 >> shift, sx1 = x1 >> shift;
    for (int sy = y0 >> shift; sy <= (y1 >> shift); ++sy) {
        if (shard_count > 1) {
            const int lo = max(y0, sy << shift), hi = min(y1, ((sy + 1) << shift) - 1);
            if (gsr_owned_rows(lo, hi, shard_index, shard_count) == 0) continue;
        }
        const uint32_t rowkey = (uint32_t)sy * (uint32_t)stiles_x;
        for (int sx = sx0; sx <= sx1; ++sx) {
            pkeys[o] = rowkey + (uint32_t)sx;
            pvals[o] = v;
            ++o;
        }
    }
}

// boundaries of equal-key runs in the sorted pair list
__global__ void __launch_bounds__(256)
k_super_ranges(const uint32_t* __restrict__ keys, uint32_t n, int32_t* __restrict__ sstart,
               int32_t* __restrict__ send)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    if (i == 0 || keys[i - 1] != k) sstart[k] = (int32_t)i;
    if (i == n - 1 || keys[i + 1] != k) send[k] = (int32_t)(i + 1);
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

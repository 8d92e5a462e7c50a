// k_sort.h -- device-wide exclusive scan and stable LSD radix sort (8-bit digits)
// of (u32 key, u32 value) pairs.
//
// Replaces tbb::parallel_sort of the index array in argsortByDistance
// (/root/reference/gsplat_plugin/src/GSplatRenderer.C:206-208) and supplies the
// (tile, depth) ordering the GL rasteriser + ordered ROP blend gave implicitly.
// Stability is what turns "sort by distance, then by tile" into per-tile
// front-to-back lists, and what makes ties deterministic (lower index first).
//
// Roofline: HBM streaming.  Per pass and item: 4 B (hist) + 8 B (scatter read)
// + 8 B (scatter write) = 20 B.  Ranking uses wave64 ballots (8 per digit
// match), per-wave LDS digit counters, and an LDS reorder so that global writes
// leave each workgroup as contiguous per-digit runs.
#pragma once
#include "gsr_device.h"

#define RS_THREADS 256
#ifndef RS_ITEMS
#define RS_ITEMS 16
#endif
#define RS_TILE (RS_THREADS * RS_ITEMS)  // items per workgroup
#define RS_WAVE_ITEMS (RS_TILE / 4)      // items per wave

// ---------------------------------------------------------------------------
// exclusive scan, three kernels (reduce / scan partials / downsweep)
#define SC_THREADS 256
#define SC_ITEMS 8
#define SC_TILE (SC_THREADS * SC_ITEMS)  // 2048

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns the
// exclusive prefix and the block total
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t t = s_wave[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ void __launch_bounds__(SC_THREADS)
k_scan_reduce(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
    uint32_t sum = 0;
    if (base + SC_ITEMS <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(in + base);
        uint4 a = p[0], b = p[1];
        sum = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    } else {
        for (int k = 0; k < SC_ITEMS; ++k)
            if (base + k < n) sum += in[base + k];
    }
    uint32_t tot;
    (void)block_excl_scan_256(sum, s_wave, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// single workgroup: in-place exclusive scan of partial[0..m), grand total -> *total
__global__ void __launch_bounds__(SC_THREADS)
k_scan_partials(uint32_t* __restrict__ partial, uint32_t m, uint32_t* __restrict__ total)
{
    __shared__ uint32_t s_wave[4];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < m; base += SC_THREADS) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = (i < m) ? partial[i] : 0u;
        uint32_t tot;
        uint32_t ex = block_excl_scan_256(v, s_wave, &tot);
        if (i < m) partial[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

__global__ void __launch_bounds__(SC_THREADS)
k_scan_down(const uint32_t* in, uint32_t n, const uint32_t* __restrict__ partial,
            uint32_t* out)  // in == out allowed (each thread reads its items before writing them)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
    uint32_t v[SC_ITEMS];
    uint32_t sum = 0;
    if (base + SC_ITEMS <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(in + base);
        uint4 a = p[0], b = p[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < SC_ITEMS; ++k) v[k] = (base + k < n) ? in[base + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) sum += v[k];
    uint32_t tot;
    uint32_t ex = block_excl_scan_256(sum, s_wave, &tot) + partial[blockIdx.x];
    uint32_t o[SC_ITEMS];
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) { o[k] = ex; ex += v[k]; }
    if (base + SC_ITEMS <= n) {
        uint4* q = reinterpret_cast<uint4*>(out + base);
        q[0] = make_uint4(o[0], o[1], o[2], o[3]);
        q[1] = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int k = 0; k < SC_ITEMS; ++k)
            if (base + k < n) out[base + k] = o[k];
    }
}

// ---------------------------------------------------------------------------
// radix pass, kernel 1: per-workgroup digit histogram -> hist[digit * nblk + blk]
__global__ void __launch_bounds__(RS_THREADS)
k_radix_hist(const uint32_t* __restrict__ keys, uint32_t n, int shift, uint32_t* __restrict__ hist, uint32_t nblk)
{
    __shared__ uint32_t h[4][256];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < 4; ++w) h[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
    if (base + RS_TILE <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(keys + base);
#pragma unroll
        for (int k = 0; k < RS_ITEMS / 4; ++k) {
            uint4 v = p[k * RS_THREADS + threadIdx.x];
            atomicAdd(&h[wave][(v.x >> shift) & 255u], 1u);
            atomicAdd(&h[wave][(v.y >> shift) & 255u], 1u);
            atomicAdd(&h[wave][(v.z >> shift) & 255u], 1u);
            atomicAdd(&h[wave][(v.w >> shift) & 255u], 1u);
        }
    } else {
        for (int k = 0; k < RS_ITEMS; ++k) {
            uint32_t i = base + k * RS_THREADS + threadIdx.x;
            if (i < n) atomicAdd(&h[wave][(keys[i] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] =
        h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
}

// radix pass, kernel 2 (after the scan of hist): stable scatter.
// Item order inside a workgroup: wave w owns items [w*RS_WAVE_ITEMS, (w+1)*RS_WAVE_ITEMS) of the
// tile, round k covers 64 consecutive items -> (wave, round, lane) is input order.
// V = payload type: uint32_t (4 B) or uint2 (8 B: splat index + packed tile rect).
template <typename V>
__global__ void __launch_bounds__(RS_THREADS)
k_radix_scatter(const uint32_t* __restrict__ keys_in, const V* __restrict__ vals_in,
                uint32_t* __restrict__ keys_out, V* __restrict__ vals_out, uint32_t n, int shift,
                const uint32_t* __restrict__ offs, uint32_t nblk)
{
    __shared__ uint32_t wc[4][256];     // per-wave digit counters -> per-wave bases
    __shared__ uint32_t dbase[256];     // first local sorted position of each digit
    __shared__ uint32_t gadj[256];      // global offset of the digit run minus dbase
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t skeys[RS_TILE];
    __shared__ V svals[RS_TILE];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile_base = blockIdx.x * RS_TILE;
    const uint32_t nvalid = (n - tile_base < RS_TILE) ? (n - tile_base) : RS_TILE;
#pragma unroll
    for (int w = 0; w < 4; ++w) wc[w][threadIdx.x] = 0;
    __syncthreads();

    uint32_t k_[RS_ITEMS], meta[RS_ITEMS];  // meta = digit | rank_in_wave_digit << 8
    V v_[RS_ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const uint32_t li = wave * RS_WAVE_ITEMS + r * 64 + lane;
        const bool valid = li < nvalid;
        uint32_t key = 0xffffffffu;
        V val{};
        if (valid) { key = keys_in[tile_base + li]; val = vals_in[tile_base + li]; }
        // invalid tail items take digit 255: being last in input order they rank
        // after every valid item and are simply not written out
        const uint32_t d = valid ? ((key >> shift) & 255u) : 255u;
        unsigned long long m = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__builtin_popcountll(m & lt_mask);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
        uint32_t prev = 0;
        if (rank == 0) {  // lowest lane of each digit group owns the counter update
            prev = wc[wave][d];
            wc[wave][d] = prev + cnt;
        }
        const int leader = __builtin_ctzll(m);
        prev = __shfl(prev, leader, 64);
        k_[r] = key; v_[r] = val;
        meta[r] = d | ((prev + rank) << 8);
    }
    __syncthreads();
    {   // thread t = digit t: wave bases, digit totals, digit bases
        const uint32_t c0 = wc[0][threadIdx.x], c1 = wc[1][threadIdx.x], c2 = wc[2][threadIdx.x],
                       c3 = wc[3][threadIdx.x];
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256(c0 + c1 + c2 + c3, s_wave, &tot);
        wc[0][threadIdx.x] = 0; wc[1][threadIdx.x] = c0; wc[2][threadIdx.x] = c0 + c1;
        wc[3][threadIdx.x] = c0 + c1 + c2;
        dbase[threadIdx.x] = ex;
        gadj[threadIdx.x] = offs[(size_t)threadIdx.x * nblk + blockIdx.x] - ex;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const uint32_t d = meta[r] & 255u;
        const uint32_t lp = dbase[d] + wc[wave][d] + (meta[r] >> 8);
        skeys[lp] = k_[r];
        svals[lp] = v_[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const uint32_t j = r * RS_THREADS + threadIdx.x;
        if (j < nvalid) {
            const uint32_t key = skeys[j];
            const uint32_t pos = gadj[(key >> shift) & 255u] + j;
            keys_out[pos] = key;
            vals_out[pos] = svals[j];
        }
    }
}

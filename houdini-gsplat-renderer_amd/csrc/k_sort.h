// k_sort.h -- the depth sort: a stable LSD radix sort (8- or 9-bit digits) of (u32 key, payload) pairs for frames that keep
// millions of splats, and -- further down -- a bucket scatter + one local kernel for the few hundred thousand an
// occlusion-culled frame keeps.
//
// Replaces tbb::parallel_sort of the index array in argsortByDistance
// (/root/reference/gsplat_plugin/src/GSplatRenderer.C:206-208) and supplies the
// (tile, depth) ordering the GL rasteriser + ordered ROP blend gave implicitly.
// Stability is what turns "sort by distance, then by tile" into per-tile
// front-to-back lists, and what makes ties deterministic (lower index first).
//
// Roofline: HBM streaming.  Per pass and item with an 8-byte payload: 4 B (hist)
// + 12 B (scatter read) + 12 B (scatter write) = 28 B.  Digits are 8 or 9 bits
// wide (9-bit digits sort a <=27-bit key range in 3 passes).  Ranking uses
// wave64 ballots (one per digit bit), per-wave LDS digit counters, and an LDS
// reorder so that global writes leave each workgroup as contiguous per-digit runs.
#pragma once
#include "gsr_device.h"

#define RS_THREADS 256
#ifndef RS_ITEMS
#define RS_ITEMS 12      // measured on C4 with XCD-contiguous tiles: 16 -> 0.205 ms, 12 -> 0.174, 8 -> 0.185, 4 -> 0.22
#endif
static_assert(RS_ITEMS % 4 == 0, "k_radix_hist reads uint4");
#ifndef RS_ITEMS_MID
#define RS_ITEMS_MID 4    // ... and for mid-size frames (k_radix_hist / k_radix_scatter's ITEMS)
#endif
#define RS_TILE (RS_THREADS * RS_ITEMS)  // items per workgroup
#define RS_WAVE_ITEMS (RS_TILE / 4)      // items per wave
#ifndef RS_XCD_DEPTH
#define RS_XCD_DEPTH 1   // depth-sort passes (scattered short runs): XCD-contiguous tiles measured 11 % faster
#endif
#ifndef RS_XCD_BIN
#define RS_XCD_BIN 0     // super-tile pass (long runs): measured slightly slower
#endif

// Workgroup b runs on XCD b % 8 (MI355X dispatch order).  An LSD pass writes, per digit, the runs of
// consecutive tiles next to each other, and a run is only a few dozen bytes -- so consecutive tiles
// should share one L2, where their partial lines merge before they are written back.  XCD x
// therefore gets the x-th CONTIGUOUS eighth of the nb tiles that exist.
__device__ __forceinline__ uint32_t rs_tile_of_block(uint32_t b, uint32_t nb, bool contig)
{
    if (!contig) return b;
    const uint32_t q = nb >> 3, r = nb & 7u, x = b & 7u, i = b >> 3;
    return x * q + (x < r ? x : r) + i;
}

// ---------------------------------------------------------------------------
// block-level scan helpers
#define SC_THREADS 256

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

// ---------------------------------------------------------------------------
// One LSD radix pass on a DBITS-wide digit (8 or 9 bits; BINS = 2^DBITS):
//   k_radix_hist<DBITS>      per-workgroup digit histogram -> hist[digit * nblk + blk]
//   k_scan_rows              one workgroup per digit: exclusive scan of its row (over workgroups),
//                            row total -> totals[digit]
//   k_radix_scatter<V,DBITS> digit bases = exclusive scan of totals (recomputed per workgroup in
//                            LDS), stable ranking, LDS reorder, coalesced runs out
// n_dev != NULL: the item count lives in device memory (it is the output of a compacting first
// pass); the grid is sized for the host-side upper bound and surplus workgroups just publish zeros.
// GATHER (the compacting first pass): the input is K1's block-compacted layout -- every RS_SRC_BLOCK slots begin with
// src_cnt[block] items, in order, and the rest of the block's slots hold nothing.  A wave takes the valid prefixes of its
// RS_WAVE_ITEMS / RS_SRC_BLOCK blocks one behind the other, so a pass over a frame that kept one splat in thirteen costs one
// round per wave instead of twelve, and the items still arrive in slot (= splat index) order.
#define RS_SRC_BLOCK 256
#define RS_WAVE_BLOCKS (RS_WAVE_ITEMS / RS_SRC_BLOCK)
static_assert(RS_WAVE_ITEMS % RS_SRC_BLOCK == 0, "a wave's share of a tile is whole K1 blocks");
// the wave's blocks' counts -> total; rs_gather_slot: q-th valid item of the wave -> its slot
template <int NB>
__device__ __forceinline__ uint32_t rs_gather_counts(const uint32_t* __restrict__ src_cnt, uint32_t first_block, uint32_t n_src_blocks,
                                                      uint32_t (&c)[NB])
{
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const uint32_t gb = first_block + (uint32_t)j;
        c[j] = gb < n_src_blocks ? src_cnt[gb] : 0u;
        tot += c[j];
    }
    return tot;
}
template <int NB>
__device__ __forceinline__ uint32_t rs_gather_slot(uint32_t q, uint32_t first_block, const uint32_t (&c)[NB])
{
    uint32_t j = 0, before = 0;
#pragma unroll
    for (int t = 0; t < NB - 1; ++t)
        if (q >= before + c[t] && j == (uint32_t)t) { before += c[t]; j = (uint32_t)t + 1u; }
    return (first_block + j) * (uint32_t)RS_SRC_BLOCK + (q - before);
}

// the digit of a pass.  CLAMP (the bucket pass of the small-frame sort, below): digit = min((key - lo) >> shift, BINS - 1), 0 for
// keys below lo -- monotone in the key, so the buckets are consecutive key ranges whatever [lo, lo + BINS << shift) turns out
// to miss
template <bool CLAMP>
__device__ __forceinline__ uint32_t rs_digit(uint32_t key, int shift, uint32_t lo, uint32_t mask)
{
    if (!CLAMP) return (key >> shift) & mask;
    const uint32_t t = key > lo ? (key - lo) >> shift : 0u;
    return t < mask ? t : mask;
}

// ITEMS = keys per thread: RS_ITEMS (12) for the frames that keep millions, RS_ITEMS_MID (4) for the ones between the small-frame sort
// and ~1.5 M keys, where a pass is bound by how long ONE workgroup takes (C3 --cull 0: +5 %, S1 +1.4 %, nothing at 4 M keys: LAB_NOTES round 5)
template <int DBITS, bool GATHER, bool CLAMP = false, int ITEMS = RS_ITEMS>
__global__ void __launch_bounds__(RS_THREADS)
k_radix_hist(const uint32_t* __restrict__ keys, uint32_t n_host, const uint32_t* __restrict__ n_dev, int shift,
             uint32_t* __restrict__ hist, uint32_t nblk, bool contig, const uint32_t* __restrict__ src_cnt, uint32_t lo = 0u)
{
    constexpr int BINS = 1 << DBITS;
    constexpr uint32_t MASK = BINS - 1;
    constexpr uint32_t TILE = RS_THREADS * ITEMS, WAVE_ITEMS = TILE / 4;
    constexpr int WAVE_BLOCKS = WAVE_ITEMS / RS_SRC_BLOCK;
    static_assert(ITEMS % 4 == 0 && WAVE_ITEMS % RS_SRC_BLOCK == 0, "uint4 reads; a wave's share of a tile is whole K1 blocks");
    __shared__ uint32_t h[4][BINS];
    const int wave = threadIdx.x >> 6;
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t nb = (n + TILE - 1) / TILE;   // tiles that exist (<= nblk, the grid's upper bound)
    if (blockIdx.x >= nb) return;                       // surplus workgroup: k_scan_rows only reads the tiles that exist
    for (int b = threadIdx.x; b < 4 * BINS; b += RS_THREADS) (&h[0][0])[b] = 0;
    __syncthreads();
    const uint32_t tile = rs_tile_of_block(blockIdx.x, nb, contig);
    const uint32_t base = tile * TILE;
    if (GATHER) {
        const uint32_t first_block = (base + (uint32_t)wave * WAVE_ITEMS) / RS_SRC_BLOCK;
        uint32_t c[WAVE_BLOCKS];
        const uint32_t tot = rs_gather_counts(src_cnt, first_block, (n + RS_SRC_BLOCK - 1) / RS_SRC_BLOCK, c);
        for (uint32_t q = threadIdx.x & 63u; q < tot; q += 64u)
            atomicAdd(&h[wave][rs_digit<CLAMP>(keys[rs_gather_slot(q, first_block, c)], shift, lo, MASK)], 1u);
    } else if (base + TILE <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(keys + base);
#pragma unroll
        for (int k = 0; k < ITEMS / 4; ++k) {
            uint4 v = p[k * RS_THREADS + threadIdx.x];
            atomicAdd(&h[wave][rs_digit<CLAMP>(v.x, shift, lo, MASK)], 1u);
            atomicAdd(&h[wave][rs_digit<CLAMP>(v.y, shift, lo, MASK)], 1u);
            atomicAdd(&h[wave][rs_digit<CLAMP>(v.z, shift, lo, MASK)], 1u);
            atomicAdd(&h[wave][rs_digit<CLAMP>(v.w, shift, lo, MASK)], 1u);
        }
    } else {
        for (int k = 0; k < ITEMS; ++k) {
            uint32_t i = base + k * RS_THREADS + threadIdx.x;
            if (i < n) atomicAdd(&h[wave][rs_digit<CLAMP>(keys[i], shift, lo, MASK)], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BINS; b += RS_THREADS)
        hist[(size_t)b * nblk + tile] = h[0][b] + h[1][b] + h[2][b] + h[3][b];
}

// grid = BINS workgroups; workgroup d scans row d of hist in place (exclusive), total -> totals[d].
// Rows are nblk entries apart; only the first `used` = ceil(items / per_block) of them hold anything (items = *n_dev when
// the count lives on the device: the grids are sized for the host-side upper bound, the surplus blocks write nothing).
__global__ void __launch_bounds__(SC_THREADS)
k_scan_rows(uint32_t* __restrict__ hist, uint32_t nblk, uint32_t* __restrict__ totals, const uint32_t* __restrict__ n_dev,
            uint32_t n_host, uint32_t per_block)
{
    __shared__ uint32_t s_wave[4];
    uint32_t* row = hist + (size_t)blockIdx.x * nblk;
    const uint32_t items = n_dev ? *n_dev : n_host;
    uint32_t used = (items + per_block - 1) / per_block;
    if (used > nblk) used = nblk;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < used; base += SC_THREADS * 4) {
        const uint32_t i0 = base + threadIdx.x * 4;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (i0 + k < used) ? row[i0 + k] : 0u;
        uint32_t tot;
        uint32_t ex = carry + block_excl_scan_256(v[0] + v[1] + v[2] + v[3], s_wave, &tot);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < used) row[i0 + k] = ex;
            ex += v[k];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// V = payload type: uint32_t (4 B) or uint2 (8 B: splat index + packed tile rect).
// Item order inside a workgroup: wave w owns items [w*RS_WAVE_ITEMS, (w+1)*RS_WAVE_ITEMS) of the
// tile, round k covers 64 consecutive items -> (wave, round, lane) is input order.
template <typename V, int DBITS, bool GATHER, bool CLAMP = false, int ITEMS = RS_ITEMS>
__global__ void __launch_bounds__(RS_THREADS)
k_radix_scatter(const uint32_t* __restrict__ keys_in, const V* __restrict__ vals_in,
                uint32_t* __restrict__ keys_out, V* __restrict__ vals_out, uint32_t n_host,
                const uint32_t* __restrict__ n_dev, int shift,
                const uint32_t* __restrict__ offs, const uint32_t* __restrict__ totals, uint32_t nblk, bool contig,
                uint32_t* __restrict__ n_out /* compacting pass: the number of surviving items (sum of the digit totals), or NULL */,
                const uint32_t* __restrict__ src_cnt /* GATHER: items at the head of every RS_SRC_BLOCK slots */, uint32_t lo = 0u)
{
    constexpr int BINS = 1 << DBITS;
    constexpr uint32_t MASK = BINS - 1;
    constexpr uint32_t TILE = RS_THREADS * ITEMS, WAVE_ITEMS = TILE / 4;
    constexpr int WAVE_BLOCKS = WAVE_ITEMS / RS_SRC_BLOCK;
    constexpr int PER = BINS / RS_THREADS;   // digits per thread in the bookkeeping step (1 or 2)
    __shared__ uint32_t wc[4][BINS];    // per-wave digit counters -> per-wave bases
    __shared__ uint32_t dbase[BINS];    // first local sorted position of each digit
    __shared__ uint32_t gadj[BINS];     // global position of the digit run minus dbase
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t skeys[TILE];
    __shared__ V svals[TILE];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t nb = (n + TILE - 1) / TILE;
    if (blockIdx.x >= nb) return;   // surplus workgroup of an upper-bound grid
    const uint32_t tile = rs_tile_of_block(blockIdx.x, nb, contig);
    const uint32_t tile_base = tile * TILE;
    const uint32_t nvalid = (n - tile_base < TILE) ? (n - tile_base) : TILE;
    for (int b = threadIdx.x; b < 4 * BINS; b += RS_THREADS) (&wc[0][0])[b] = 0;
    // digit bases: exclusive scan of the per-digit totals (thread t owns digits t*PER .. t*PER+PER-1)
    {
        uint32_t tv[PER], tsum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { tv[k] = totals[threadIdx.x * PER + k]; tsum += tv[k]; }
        uint32_t tot;
        uint32_t ex = block_excl_scan_256(tsum, s_wave, &tot);   // (contains the __syncthreads for wc too)
#pragma unroll
        for (int k = 0; k < PER; ++k) { gadj[threadIdx.x * PER + k] = ex; ex += tv[k]; }
        if (n_out && blockIdx.x == 0 && threadIdx.x == 0) *n_out = tot;   // read by the next pass's kernels
    }
    __syncthreads();

    uint32_t k_[ITEMS], meta[ITEMS];  // meta = digit | rank_in_wave_digit << DBITS, 0xffffffff = no item
    __shared__ uint32_t s_tile_items;
    V v_[ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t gc[WAVE_BLOCKS];
    const uint32_t g_first = (tile_base + (uint32_t)wave * WAVE_ITEMS) / RS_SRC_BLOCK;
    const uint32_t g_tot = GATHER ? rs_gather_counts(src_cnt, g_first, (n + RS_SRC_BLOCK - 1) / RS_SRC_BLOCK, gc) : 0u;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t li = wave * WAVE_ITEMS + r * 64 + lane;
        bool valid = li < nvalid;
        uint32_t key = 0xffffffffu;
        V val{};
        if (GATHER) {   // compacting pass: the valid prefixes of the wave's source blocks, one behind the other
            if ((uint32_t)(r * 64) >= g_tot) { k_[r] = key; v_[r] = val; meta[r] = 0xffffffffu; continue; }   // (wave-uniform)
            const uint32_t q = (uint32_t)(r * 64 + lane);
            valid = q < g_tot;
            if (valid) {
                const uint32_t slot = rs_gather_slot(q, g_first, gc);
                key = keys_in[slot];
                val = vals_in[slot];
            }
        } else if (valid) {
            key = keys_in[tile_base + li];
            val = vals_in[tile_base + li];
        }
        // items that do not exist neither rank nor count nor get written
        const uint32_t d = rs_digit<CLAMP>(key, shift, lo, MASK);
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < DBITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__builtin_popcountll(m & lt_mask);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
        uint32_t prev = 0;
        if (valid && rank == 0) {  // lowest lane of each digit group owns the counter update
            prev = wc[wave][d];
            wc[wave][d] = prev + cnt;
        }
        const int leader = m ? __builtin_ctzll(m) : 0;
        prev = __shfl(prev, leader, 64);
        k_[r] = key; v_[r] = val;
        meta[r] = valid ? (d | ((prev + rank) << DBITS)) : 0xffffffffu;
    }
    __syncthreads();
    {   // thread t owns digits t*PER..: wave bases, digit totals in this tile, local digit bases
        uint32_t tot_d[PER], tsum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int d = threadIdx.x * PER + k;
            const uint32_t c0 = wc[0][d], c1 = wc[1][d], c2 = wc[2][d], c3 = wc[3][d];
            wc[0][d] = 0; wc[1][d] = c0; wc[2][d] = c0 + c1; wc[3][d] = c0 + c1 + c2;
            tot_d[k] = c0 + c1 + c2 + c3;
            tsum += tot_d[k];
        }
        uint32_t tot;
        uint32_t ex = block_excl_scan_256(tsum, s_wave, &tot);
        if (threadIdx.x == 0) s_tile_items = tot;   // items of this tile that exist
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int d = threadIdx.x * PER + k;
            dbase[d] = ex;
            // global position of this tile's run of digit d = digit base + rank of the tile inside the digit
            gadj[d] = gadj[d] + offs[(size_t)d * nblk + tile] - ex;
            ex += tot_d[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        if (meta[r] == 0xffffffffu) continue;
        const uint32_t d = meta[r] & MASK;
        const uint32_t lp = dbase[d] + wc[wave][d] + (meta[r] >> DBITS);
        skeys[lp] = k_[r];
        svals[lp] = v_[r];
    }
    __syncthreads();
    const uint32_t tile_items = s_tile_items;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t j = r * RS_THREADS + threadIdx.x;
        if (j < tile_items) {
            const uint32_t key = skeys[j];
            const uint32_t pos = gadj[rs_digit<CLAMP>(key, shift, lo, MASK)] + j;
            keys_out[pos] = key;
            vals_out[pos] = svals[j];
        }
    }
}

// ---------------------------------------------------------------------------
// Position-keyed order (GSR_OPT_SORT_CACHE = 2; the reference's rule: argsortByDistance runs only when the camera POSITION moves,
// /root/reference/gsplat_plugin/src/GSplatRenderer.C:165-186).  The distance keys depend on the position alone, so ALL splats are
// sorted once per position (k_pos_keys + the LSD passes above); while the position stands still K1 walks the splats in that
// order, its per-workgroup compaction leaves what a frame keeps already sorted, and two small kernels make it dense.
__global__ void __launch_bounds__(256)
k_pos_keys(const float4* __restrict__ geoA, uint32_t n, float cx, float cy, float cz, uint32_t key_min, uint32_t key_max,
           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = geoA[i];
    // distance^2 with the operations of k_preprocess.h (gsr_k1_front) -- the same key bits
    const float dx = a.x - cx, dy = a.y - cy, dz = a.z - cz;
    const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    uint32_t b = __builtin_bit_cast(uint32_t, d2);
    b = b < key_min ? key_min : (b > key_max ? key_max : b);
    keys[i] = b - key_min;
    vals[i] = i;
}
// exclusive scan of cnt[0, m) (m = *slots_dev / RS_SRC_BLOCK workgroup-iterations of K1) -> pre[0, m); the total -> *n_out.  One workgroup.
__global__ void __launch_bounds__(1024)
k_scan_counts(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ slots_dev, uint32_t* __restrict__ pre, uint32_t* __restrict__ n_out)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const uint32_t m = *slots_dev / (uint32_t)RS_SRC_BLOCK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < m; base += 1024u) {
        const uint32_t j = base + threadIdx.x;
        const uint32_t v = j < m ? cnt[j] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_w[w];
        const uint32_t carry = s_carry;
        if (j < m) pre[j] = carry + wbase + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + wbase + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = s_carry;
}
// K1's per-workgroup-iteration compaction (cnt[k] items at the head of slots [256 k, 256 k + 256)) -> dense arrays
template <typename V>
__global__ void __launch_bounds__(RS_SRC_BLOCK)
k_compact_blocks(const uint32_t* __restrict__ keys_in, const V* __restrict__ vals_in, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ pre,
                 const uint32_t* __restrict__ slots_dev, uint32_t* __restrict__ keys_out, V* __restrict__ vals_out)
{
    const uint32_t k = blockIdx.x;
    if (k * (uint32_t)RS_SRC_BLOCK >= *slots_dev) return;
    if (threadIdx.x < cnt[k]) {
        const uint32_t src = k * (uint32_t)RS_SRC_BLOCK + threadIdx.x, dst = pre[k] + threadIdx.x;
        keys_out[dst] = keys_in[src];
        vals_out[dst] = vals_in[src];
    }
}

// ---------------------------------------------------------------------------
// Small sorts: ONE bucket kernel + ONE local kernel.
// After occlusion culling a frame sorts a few hundred thousand keys, and the three LSD passes above are nine launches at
// their latency floors (65 us for 0.3 M keys on MI355X).  Instead:
//   k_bucket_scatter  drops every key into one of BK_BUCKETS (1024) buckets of equal width over the key range the slot's PREVIOUS frame kept
//                     (+ margins; what falls outside goes to the first / last bucket).  Every bucket owns a fixed region of
//                     BK_CAP slots; a workgroup counts its keys per bucket in LDS and reserves room with one atomic per bucket
//                     (counters 256 bytes apart: atomics that share a cache line serialise) -- no histogram / scan launches;
//   k_radix_local     one workgroup per bucket sorts it on the bits the bucket index does not fix -- LSD passes of 8 bits that
//                     never leave the CU for up to RL_CHUNK keys (registers + LDS), chunked through global memory beyond --
//                     and writes it to its place in the dense output (exclusive prefix of the bucket counts).
// The order INSIDE a bucket after the scatter is whatever the atomics made it, so the contract's tie order (equal keys in
// storage order) is restored explicitly: runs of equal keys are re-ordered by their payload's splat index.
// Correct for ANY prediction; what it cannot do in reasonable time -- a bucket that overflows its region, an endless run of
// equal keys -- raises *failed, and the host renders the frame again with the three global passes (gsr_api.hip).
#define RL_THREADS 256
#ifndef RL_ITEMS
#define RL_ITEMS 8
#endif
#define RL_CHUNK (RL_THREADS * RL_ITEMS)     // 2048 items: 8 KB of keys + 16 KB of payloads in LDS, 86 VGPRs
#define RL_WAVE_ITEMS (RL_CHUNK / 4)
#define RL_BINS 256
static_assert(BK_CAP >= RL_CHUNK, "a bucket's region holds at least one chunk");
#define RL_MAX_RUN 64                        // longest run of equal keys re-ordered in place

__device__ __forceinline__ uint32_t rl_id(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t rl_id(uint2 v) { return v.x; }

template <typename V, bool GATHER>
__global__ void __launch_bounds__(RS_THREADS)
k_bucket_scatter(const uint32_t* __restrict__ keys_in, const V* __restrict__ vals_in, uint32_t n_host, const uint32_t* __restrict__ n_dev,
                 int shift, uint32_t lo, const uint32_t* __restrict__ src_cnt /* GATHER: items at the head of every RS_SRC_BLOCK slots */,
                 uint32_t* __restrict__ gcnt /* [BK_BUCKETS * BK_STRIDE], zero on entry */, uint32_t* __restrict__ kout, V* __restrict__ vout,
                 uint32_t* __restrict__ n_out /* += keys scattered (zero on entry), or NULL */, uint32_t* __restrict__ failed)
{
    __shared__ uint32_t h[BK_BUCKETS];
    __shared__ uint32_t s_wave[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    KPROF(1, 0)
    KPROF_BLK_BEGIN
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t nb = (n + RS_TILE - 1) / RS_TILE;
    if (blockIdx.x >= nb) return;
    for (int b = threadIdx.x; b < BK_BUCKETS; b += RS_THREADS) h[b] = 0;
    __syncthreads();
    KPROF(1, 1)
    const uint32_t tile_base = blockIdx.x * RS_TILE;
    const uint32_t nvalid = (n - tile_base < RS_TILE) ? (n - tile_base) : RS_TILE;
    uint32_t k_[RS_ITEMS], meta[RS_ITEMS];   // meta = bucket | rank in (workgroup, bucket) << 12, 0xffffffff = no item
    static_assert(BK_BUCKETS <= 4096 && RS_TILE < (1 << 20), "meta packing");
    V v_[RS_ITEMS];
    uint32_t gc[RS_WAVE_BLOCKS];
    const uint32_t g_first = (tile_base + (uint32_t)wave * RS_WAVE_ITEMS) / RS_SRC_BLOCK;
    const uint32_t g_tot = GATHER ? rs_gather_counts(src_cnt, g_first, (n + RS_SRC_BLOCK - 1) / RS_SRC_BLOCK, gc) : 0u;
    uint32_t mine = 0;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const uint32_t li = wave * RS_WAVE_ITEMS + r * 64 + lane;
        bool valid = li < nvalid;
        uint32_t key = 0u;
        V val{};
        if (GATHER) {
            const uint32_t q = (uint32_t)(r * 64 + lane);
            valid = q < g_tot;
            if (valid) { const uint32_t slot = rs_gather_slot(q, g_first, gc); key = keys_in[slot]; val = vals_in[slot]; }
        } else if (valid) {
            key = keys_in[tile_base + li]; val = vals_in[tile_base + li];
        }
        k_[r] = key; v_[r] = val; meta[r] = 0xffffffffu;
        if (valid) {
            const uint32_t d = rs_digit<true>(key, shift, lo, BK_BUCKETS - 1);
            meta[r] = d | (atomicAdd(&h[d], 1u) << 12);
            ++mine;
        }
    }
    __syncthreads();
    KPROF(1, 2)
    for (int d = threadIdx.x; d < BK_BUCKETS; d += RS_THREADS) {   // one reservation per bucket this workgroup has keys for
        const uint32_t c = h[d];
        h[d] = c ? atomicAdd(&gcnt[(size_t)d * BK_STRIDE], c) : 0u;
    }
    if (n_out) {
        uint32_t tot;
        (void)block_excl_scan_256(mine, s_wave, &tot);
        if (threadIdx.x == 0 && tot) atomicAdd(n_out, tot);
    }
    __syncthreads();
    KPROF(1, 3)
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        if (meta[r] == 0xffffffffu) continue;
        const uint32_t d = meta[r] & 4095u, p = h[d] + (meta[r] >> 12);
        if (p < (uint32_t)BK_CAP) { kout[(size_t)d * BK_CAP + p] = k_[r]; vout[(size_t)d * BK_CAP + p] = v_[r]; }
        else if (failed) *failed = 1u;            // the bucket's region is full: the prediction missed badly
    }
    KPROF(1, 4)
    KPROF_BLK_END(1, mine)
}

// The scatter for K1's output as it is: one WAVE per K1 workgroup-iteration (whose kept splats sit at the head of its 256 slots,
// blk_cnt of them), four of them per workgroup -- half the slots of k_bucket_scatter's workgroups, twice the workgroups, no
// gather arithmetic: a frame of a few hundred thousand keys is bound by how long ONE workgroup takes, not by throughput.
// The total is left to k_radix_local.
template <typename V>
__global__ void __launch_bounds__(256)
k_bucket_scatter_k1(const uint32_t* __restrict__ keys_in, const V* __restrict__ vals_in, uint32_t n_blocks_host, const uint32_t* __restrict__ n_dev,
                    int shift, uint32_t lo, const uint32_t* __restrict__ src_cnt, uint32_t* __restrict__ gcnt,
                    uint32_t* __restrict__ kout, V* __restrict__ vout, uint32_t* __restrict__ failed,
                    const uint32_t* __restrict__ range_dev = nullptr /* or: { lo, shift } on the device (front-slab phases: k_slab_pick) */)
{
    __shared__ uint32_t h[BK_BUCKETS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (range_dev) { lo = range_dev[0]; shift = (int)range_dev[1]; }
    // The grid follows the slot's previous frame (like K1's): a frame that fills more slots loops.  The first trip's loads are
    // requested together with the slot count that says whether they exist (every address does: clamped to the host's bound).
    uint32_t n = 0xffffffffu;
    for (uint32_t wg = blockIdx.x; wg * 4u * (uint32_t)RS_SRC_BLOCK < n; wg += gridDim.x) {
        uint32_t gb = wg * 4u + (uint32_t)wave;                        // this wave's K1 block
        const bool surplus = gb >= n_blocks_host;
        gb = surplus ? n_blocks_host - 1u : gb;
        const uint32_t base = gb * (uint32_t)RS_SRC_BLOCK;
        const uint32_t cnt_raw = src_cnt[gb];
        uint32_t k_[4];
        V v_[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { k_[r] = keys_in[base + (uint32_t)(r * 64 + lane)]; v_[r] = vals_in[base + (uint32_t)(r * 64 + lane)]; }
        if (n == 0xffffffffu) {
            n = *n_dev;
            if (wg * 4u * (uint32_t)RS_SRC_BLOCK >= n) return;
        }
        const uint32_t cnt = (base < n && !surplus) ? cnt_raw : 0u;
        for (int b = threadIdx.x; b < BK_BUCKETS; b += 256) h[b] = 0;
        __syncthreads();
        uint32_t meta[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            meta[r] = 0xffffffffu;
            if ((uint32_t)(r * 64 + lane) < cnt) {
                const uint32_t d = rs_digit<true>(k_[r], shift, lo, BK_BUCKETS - 1);
                meta[r] = d | (atomicAdd(&h[d], 1u) << 12);
            }
        }
        __syncthreads();
        for (int d = threadIdx.x; d < BK_BUCKETS; d += 256) {   // one reservation per bucket this workgroup has keys for
            const uint32_t c = h[d];
            h[d] = c ? atomicAdd(&gcnt[(size_t)d * BK_STRIDE], c) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (meta[r] == 0xffffffffu) continue;
            const uint32_t d = meta[r] & 4095u, p = h[d] + (meta[r] >> 12);
            if (p < (uint32_t)BK_CAP) { kout[(size_t)d * BK_CAP + p] = k_[r]; vout[(size_t)d * BK_CAP + p] = v_[r]; }
            else if (failed) *failed = 1u;
        }
        __syncthreads();   // (the next trip clears the counts)
    }
}

// The same scatter for the frames this sort is made for, one workgroup per K1 workgroup-iteration (its kept splats sit at the
// head of its 256 slots): a key's place in its bucket is simply what one global atomic on the bucket's counter returns.  No LDS,
// no barrier, three dependent trips to memory -- and ten times the workgroups: k_bucket_scatter's 2048-slot workgroups leave
// a frame of a few hundred thousand keys (let alone ten thousand) on a handful of CUs.  The atomics are about as many either
// way (a workgroup of k_bucket_scatter finds its ~900 keys in ~600 different buckets).  The total is left to k_radix_local.
template <typename V>
__global__ void __launch_bounds__(RS_SRC_BLOCK)
k_bucket_scatter_direct(const uint32_t* __restrict__ keys_in, const V* __restrict__ vals_in, const uint32_t* __restrict__ n_dev,
                        int shift, uint32_t lo, const uint32_t* __restrict__ src_cnt, uint32_t* __restrict__ gcnt,
                        uint32_t* __restrict__ kout, V* __restrict__ vout, uint32_t* __restrict__ failed)
{
    const uint32_t slot = blockIdx.x * (uint32_t)RS_SRC_BLOCK + threadIdx.x;
    // (requested together; the grid covers the slots K1 can fill at most, so both addresses exist)
    const uint32_t cnt = src_cnt[blockIdx.x];
    const uint32_t key = keys_in[slot];
    const V val = vals_in[slot];
    const uint32_t n = *n_dev;
    if (blockIdx.x * (uint32_t)RS_SRC_BLOCK >= n || threadIdx.x >= cnt) return;
    const uint32_t d = rs_digit<true>(key, shift, lo, BK_BUCKETS - 1);
    const uint32_t p = atomicAdd(&gcnt[(size_t)d * BK_STRIDE], 1u);
    if (p < (uint32_t)BK_CAP) { kout[(size_t)d * BK_CAP + p] = key; vout[(size_t)d * BK_CAP + p] = val; }
    else if (failed) *failed = 1u;
}

// one stable 8-bit pass over the n (<= RL_CHUNK) items a workgroup holds in registers in (wave, round, lane) order:
// on return skeys / svals hold them sorted by the digit, and dcount[d] = items with digit d, dbase[d] = their first position
// wq = items per wave (a multiple of 64, <= RL_WAVE_ITEMS): wave w holds items [w * wq, (w + 1) * wq) -- a small bucket is
// spread over the four waves instead of filling the first one round after round
template <typename V>
__device__ __forceinline__ void rl_pass_in_lds(uint32_t (&k_)[RL_ITEMS], V (&v_)[RL_ITEMS], uint32_t n, uint32_t wq, int shift, uint32_t sub,
                                               uint32_t (*wc)[RL_BINS], uint32_t* dbase, uint32_t* dcount, uint32_t* s_wave,
                                               uint32_t* skeys, V* svals)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int b = threadIdx.x; b < 4 * RL_BINS; b += RL_THREADS) (&wc[0][0])[b] = 0;
    __syncthreads();
    uint32_t meta[RL_ITEMS];
#pragma unroll
    for (int r = 0; r < RL_ITEMS; ++r) {
        const uint32_t li = (uint32_t)wave * wq + (uint32_t)r * 64u + (uint32_t)lane;
        const bool valid = (uint32_t)r * 64u < wq && li < n;
        const uint32_t d = ((k_[r] - sub) >> shift) & (RL_BINS - 1);
        unsigned long long m = __ballot(valid);
        if (m == 0ull) { meta[r] = 0xffffffffu; continue; }   // (wave-uniform)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__builtin_popcountll(m & lt_mask);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
        uint32_t prev = 0;
        if (valid && rank == 0) { prev = wc[wave][d]; wc[wave][d] = prev + cnt; }
        const int leader = m ? __builtin_ctzll(m) : 0;
        prev = __shfl(prev, leader, 64);
        meta[r] = valid ? (d | ((prev + rank) << 8)) : 0xffffffffu;
    }
    __syncthreads();
    {   // thread t owns digit t: wave bases, the digit's count and first sorted position
        const int d = threadIdx.x;
        const uint32_t c0 = wc[0][d], c1 = wc[1][d], c2 = wc[2][d], c3 = wc[3][d];
        wc[0][d] = 0; wc[1][d] = c0; wc[2][d] = c0 + c1; wc[3][d] = c0 + c1 + c2;
        const uint32_t tot_d = c0 + c1 + c2 + c3;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256(tot_d, s_wave, &tot);
        dbase[d] = ex;
        dcount[d] = tot_d;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RL_ITEMS; ++r) {
        if (meta[r] == 0xffffffffu) continue;
        const uint32_t d = meta[r] & (RL_BINS - 1);
        const uint32_t lp = dbase[d] + wc[wave][d] + (meta[r] >> 8);
        skeys[lp] = k_[r];
        svals[lp] = v_[r];
    }
    __syncthreads();
}

// equal keys in storage order: every run of equal keys in keys[0, n) (sorted) is re-ordered by its payloads' splat index.
// One thread per run (they are rare and short); a run beyond RL_MAX_RUN raises *failed instead.  All threads; ends in a barrier.
template <typename V>
__device__ __forceinline__ void rl_fix_ties(uint32_t* keys, V* vals, uint32_t n, uint32_t* failed)
{
    for (uint32_t j = threadIdx.x; j + 1u < n; j += RL_THREADS) {
        const uint32_t key = keys[j];
        if (keys[j + 1u] != key || (j > 0u && keys[j - 1u] == key)) continue;    // not the start of a run
        uint32_t e = j + 2u;
        while (e < n && keys[e] == key) ++e;
        if (failed && e - j > (uint32_t)RL_MAX_RUN) { *failed = 1u; continue; }   // (failed == NULL: whatever it takes)
        for (uint32_t a = j + 1u; a < e; ++a) {          // insertion sort of vals[j, e) by splat index
            const V x = vals[a];
            uint32_t b = a;
            while (b > j && rl_id(vals[b - 1u]) > rl_id(x)) { vals[b] = vals[b - 1u]; --b; }
            vals[b] = x;
        }
    }
    __syncthreads();
}

// grid = BK_BUCKETS.  Bucket b = the cnt[b * BK_STRIDE] keys at src + b * BK_CAP (k_bucket_scatter); it is sorted and written to
// dst at the exclusive prefix of the counts.  A middle bucket holds keys of [lo + b << low_bits, lo + (b + 1) << low_bits): sorted
// on the low_bits low bits of key - lo; the first and the last bucket also hold whatever fell outside the predicted range: all
// full_bits bits.  failed == NULL (tests): long runs of equal keys are re-ordered whatever it takes.
template <typename V>
__global__ void __launch_bounds__(RL_THREADS)
k_radix_local(const uint32_t* __restrict__ cnt, int low_bits, int full_bits, uint32_t lo,
              uint32_t* __restrict__ ksrc, V* __restrict__ vsrc, uint32_t* __restrict__ kdst, V* __restrict__ vdst,
              uint32_t* __restrict__ failed, uint32_t* __restrict__ n_out /* the last workgroup leaves the number of keys here (or NULL) */,
              const uint32_t* __restrict__ range_dev = nullptr /* or: { lo, low_bits } on the device (front-slab phases: k_slab_pick) */)
{
    if (range_dev) { lo = range_dev[0]; low_bits = (int)range_dev[1]; }
    __shared__ uint32_t wc[4][RL_BINS];
    __shared__ uint32_t dbase[RL_BINS], dcount[RL_BINS], gbase[RL_BINS];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t skeys[RL_CHUNK];
    __shared__ V svals[RL_CHUNK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = (int)blockIdx.x;
    KPROFB(2, 0, BK_BUCKETS / 2)
    KPROF_BLK_BEGIN
    // where the bucket goes: the (clamped) counts of the buckets before it
    uint32_t before = 0;
    for (int d = threadIdx.x; d < b; d += RL_THREADS) { const uint32_t c = cnt[(size_t)d * BK_STRIDE]; before += c < (uint32_t)BK_CAP ? c : (uint32_t)BK_CAP; }
    uint32_t tot;
    (void)block_excl_scan_256(before, s_wave, &tot);
    const uint32_t start = tot;
    KPROFB(2, 1, BK_BUCKETS / 2)
    uint32_t n = cnt[(size_t)b * BK_STRIDE];
    n = n < (uint32_t)BK_CAP ? n : (uint32_t)BK_CAP;
    if (n_out && b == BK_BUCKETS - 1 && threadIdx.x == 0) *n_out = start + n;
    if (n == 0u) return;
    uint32_t* ks = ksrc + (size_t)b * BK_CAP;
    V* vs = vsrc + (size_t)b * BK_CAP;
    uint32_t* kd = kdst + start;
    V* vd = vdst + start;
    const bool middle = b > 0 && b < BK_BUCKETS - 1;
    const int sort_bits = middle ? low_bits : full_bits;
    const uint32_t sub = middle ? lo : 0u;
    const int npass = sort_bits <= 0 ? 0 : (sort_bits + 7) / 8;
    uint32_t k_[RL_ITEMS];
    V v_[RL_ITEMS];
    if (n <= 64u) {
        // a handful of keys (every bucket of a frame of ten thousand splats): every thread counts the keys
        // that come before its own -- (key, splat index) pairs, so the contract's tie order falls out -- and stores its key there.
        // n broadcast reads from LDS instead of two LSD passes with six barriers each: 8.6 -> 6.8 us on BASELINE C1.  (Not beyond:
        // at 100 - 256 keys the four waves, alone on their SIMDs, take longer over the comparisons than the passes take.)
        unsigned long long* packed = reinterpret_cast<unsigned long long*>(svals);
        static_assert(sizeof(V) * RL_CHUNK >= 8 * RL_THREADS, "the payload staging area holds one packed pair per thread");
        const uint32_t t = threadIdx.x;
        uint32_t key = 0u;
        V val{};
        if (t < n) { key = ks[t]; val = vs[t]; }
        const unsigned long long me = ((unsigned long long)key << 32) | (unsigned long long)rl_id(val);
        packed[t] = t < n ? me : ~0ull;           // (the padding never counts: nothing is below it, and it equals no pair)
        __syncthreads();
        uint32_t rank = 0;
        for (uint32_t j0 = 0; j0 < n; j0 += 8u) {  // eight reads in flight: one at a time the loop is an LDS latency per key
            unsigned long long o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = packed[j0 + (uint32_t)u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += (o[u] < me || (o[u] == me && j0 + (uint32_t)u < t)) ? 1u : 0u;
        }
        if (t < n) { kd[rank] = key; vd[rank] = val; }
        KPROF_BLK_END(2, n)
        return;
    }
    if (n <= (uint32_t)RL_CHUNK) {
        // the whole bucket lives in registers + LDS for all passes, a quarter (rounded up to whole rounds of 64) per wave
        const uint32_t wq = ((n + 255u) / 256u) * 64u;
        auto mine = [&](int r, uint32_t& li) { li = (uint32_t)wave * wq + (uint32_t)r * 64u + (uint32_t)lane; return (uint32_t)r * 64u < wq && li < n; };
#pragma unroll
        for (int r = 0; r < RL_ITEMS; ++r) {
            uint32_t li;
            k_[r] = 0u; v_[r] = V{};
            if (mine(r, li)) { k_[r] = ks[li]; v_[r] = vs[li]; }
        }
        KPROFB(2, 2, BK_BUCKETS / 2)
        if (npass == 0) {      // (a bucket one key wide: nothing to sort but the ties)
#pragma unroll
            for (int r = 0; r < RL_ITEMS; ++r) {
                uint32_t li;
                if (mine(r, li)) { skeys[li] = k_[r]; svals[li] = v_[r]; }
            }
            __syncthreads();
        }
        for (int p = 0; p < npass; ++p) {
            rl_pass_in_lds(k_, v_, n, wq, 8 * p, sub, wc, dbase, dcount, s_wave, skeys, svals);
            if (p + 1 < npass) {
#pragma unroll
                for (int r = 0; r < RL_ITEMS; ++r) {
                    uint32_t li;
                    if (mine(r, li)) { k_[r] = skeys[li]; v_[r] = svals[li]; }
                }
                __syncthreads();
            }
        }
        KPROFB(2, 3, BK_BUCKETS / 2)
        rl_fix_ties(skeys, svals, n, failed);
        KPROFB(2, 4, BK_BUCKETS / 2)
        for (uint32_t j = threadIdx.x; j < n; j += RL_THREADS) { kd[j] = skeys[j]; vd[j] = svals[j]; }
        KPROFB(2, 5, BK_BUCKETS / 2)
        KPROF_BLK_END(2, n)
        return;
    }
    // a bucket larger than one chunk: every pass goes through global memory, chunk by chunk in order (src -> dst -> src ...)
    uint32_t* ka = ks; V* va = vs;
    uint32_t* kb = kd; V* vb = vd;
    for (int p = 0; p < npass; ++p) {
        const int shift = 8 * p;
        // digit counts of the whole bucket -> first position of every digit
        for (int d = threadIdx.x; d < 4 * RL_BINS; d += RL_THREADS) (&wc[0][0])[d] = 0;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < n; j += RL_THREADS) atomicAdd(&wc[wave][((ka[j] - sub) >> shift) & (RL_BINS - 1)], 1u);
        __syncthreads();
        {
            const int d = threadIdx.x;
            const uint32_t c = wc[0][d] + wc[1][d] + wc[2][d] + wc[3][d];
            uint32_t t2;
            gbase[d] = block_excl_scan_256(c, s_wave, &t2);
        }
        __syncthreads();
        for (uint32_t c0 = 0; c0 < n; c0 += RL_CHUNK) {
            const uint32_t m = n - c0 < (uint32_t)RL_CHUNK ? n - c0 : (uint32_t)RL_CHUNK;
#pragma unroll
            for (int r = 0; r < RL_ITEMS; ++r) {
                const uint32_t li = (uint32_t)wave * RL_WAVE_ITEMS + (uint32_t)r * 64u + (uint32_t)lane;
                k_[r] = 0u; v_[r] = V{};
                if (li < m) { k_[r] = ka[c0 + li]; v_[r] = va[c0 + li]; }
            }
            rl_pass_in_lds(k_, v_, m, (uint32_t)RL_WAVE_ITEMS, shift, sub, wc, dbase, dcount, s_wave, skeys, svals);
            for (uint32_t j = threadIdx.x; j < m; j += RL_THREADS) {
                const uint32_t key = skeys[j];
                const uint32_t d = ((key - sub) >> shift) & (RL_BINS - 1);
                const uint32_t pos = gbase[d] + (j - dbase[d]);
                kb[pos] = key;
                vb[pos] = svals[j];
            }
            __syncthreads();
            gbase[threadIdx.x] += dcount[threadIdx.x];     // (thread t owns digit t)
            __syncthreads();
        }
        __threadfence();    // this workgroup re-reads what it wrote (other lanes' stores) in the next pass
        __syncthreads();
        uint32_t* tk = ka; ka = kb; kb = tk;
        V* tv = va; va = vb; vb = tv;
    }
    if (ka != kd) {   // an even number of passes left the bucket in its region of src: copy it over
        for (uint32_t j = threadIdx.x; j < n; j += RL_THREADS) { kd[j] = ka[j]; vd[j] = va[j]; }
        __threadfence();
        __syncthreads();
    }
    rl_fix_ties(kd, vd, n, failed);
    KPROF_BLK_END(2, n)
}

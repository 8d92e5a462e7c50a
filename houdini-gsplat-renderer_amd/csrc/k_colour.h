// k_colour.h -- LAZY colour: SH is evaluated only for the splats a frame can actually composite.
//
// The reference evaluates ShadeSH for every splat (six times: once per quad corner,
// /root/reference/gsplat_plugin/shaders/GSplatShaderSource.h:244-274).  With front-to-back compositing and a
// per-pixel early-out, a tile stops after the first few percent of its super-tile's depth-ordered list -- on the
// 6 M-splat BASELINE scene 4.35 M splats are visible but fewer than 0.2 M are ever composited -- so K1 leaves the
// colour of a record PENDING (GSR_COLOUR_PENDING in its `r`) and
//   k_colour_prefix   evaluates it for the FRONT of every super-tile list: as many entries as any tile of that
//                     super-tile scanned in the previous frame, +25 % + 1024 (k_sum_work keeps the figure);
//   k_blend           bails out of a tile that meets a record that is still pending (the prediction fell short:
//                     first frame, a camera jump) and appends the tile to a redo list;
//   k_blend<.., LAZY> composites the redo tiles with on-demand colour evaluation in the gather.
// Colours come from the one function gsr_splat_colour() in all three places: pixels are bit-identical to eager
// evaluation (GSR_OPT_LAZY_COLOUR = 0), which tests/test_gpu_parity.py checks.
#pragma once
#include "gsr_device.h"
#include "k_preprocess.h"

#define CL_THREADS 256
#ifndef CL_BLOCKS_PER_LIST
#define CL_BLOCKS_PER_LIST 8      // workgroups striding over one super-tile's list prefix (measured on C4: 8 -> 54 us, 16 -> 61, 32 -> 83)
#endif

// grid = n_super * CL_BLOCKS_PER_LIST.  prefix[s] = entries of list s to colour (0xffffffff = all; written by k_sum_work).
// A splat sits in every super-tile list it reaches (1.8 on average) and is simply evaluated once per list: the
// evaluations agree bit for bit, so the racing stores are harmless -- and any "first one wins" protocol needs a global
// atomic per splat, which MI355X executes at ~6 G/s (measured: 0.4 M atomicOr/atomicCAS = 67 us, against 13 us for the
// 0.4 M evaluations themselves).
// Memory pipeline: a lane reading its own 128-byte row piece by piece makes every load instruction touch 64 different
// lines (PMC: 76 % of the wave cycles were issue stalls behind the vector-memory unit).  So the rows of a wave's 64
// splats are fetched COOPERATIVELY -- eight lanes per row, eight rows per load instruction, each instruction eight full
// lines -- into LDS, each lane then reads its own row from there, and the result goes out as ONE 16-byte store.
#define CL_ROW_DW 36   // LDS row pitch in dwords (32 + 4: neighbouring lanes' rows start in different banks)
// colours of list entries e = first, first + step, ... < hi (wave-uniform bounds; the wave takes 64 consecutive entries per trip)
__device__ __forceinline__ uint32_t
cl_colour_span(const GsrFrame& f, const uint2* __restrict__ lists, int first, int hi, int step, const uint4* __restrict__ colrow,
               GsrRecord* __restrict__ rec, uint32_t* srow, uint32_t* sidx)
{
    const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
    uint32_t mine = 0;
    for (int e0 = first; e0 < hi; e0 += step) {   // (wave-uniform trip count)
        const int e = e0 + lane;
        const bool live = e < hi;
        const uint32_t idx = live ? (lists[e].x & f.idx_mask) : 0xffffffffu;
        sidx[wbase + lane] = idx;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // lane l fetches piece (l & 7) of the rows of splats (l >> 3) + 8 k, k = 0..7, of this wave
        uint4 piece[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t j = sidx[wbase + k * 8 + (lane >> 3)];
            piece[k] = (j != 0xffffffffu) ? colrow[(size_t)j * 8 + (lane & 7)] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            *reinterpret_cast<uint4*>(&srow[(wbase + k * 8 + (lane >> 3)) * CL_ROW_DW + (lane & 7) * 4]) = piece[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (live) {
            const uint4* row = reinterpret_cast<const uint4*>(&srow[(wbase + lane) * CL_ROW_DW]);
            float cr, cg, cb;
            gsr_splat_colour_from_row(f, row, 0u, cr, cg, cb);
            const float opacity = __uint_as_float(row[0].w);
            reinterpret_cast<float4*>(rec + idx)[2] = make_float4(cr, cg, cb, gsr_log2_opacity(opacity));   // (r, g, b, la): the record's third quad
            ++mine;
        }
        __builtin_amdgcn_wave_barrier();   // the rows are overwritten by the next trip
    }
    return mine;
}

__global__ void __launch_bounds__(CL_THREADS)
k_colour_prefix(GsrFrame f, const uint2* __restrict__ lists, const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
                const uint32_t* __restrict__ prefix, int list_cap, const uint4* __restrict__ colrow /* 8 x 16 B per splat */,
                GsrRecord* __restrict__ rec,
                uint32_t* __restrict__ evals /* [256] colours evaluated per super-tile list this frame (diagnostics; summed by k_sum_work) */)
{
    __shared__ uint32_t srow[CL_THREADS * CL_ROW_DW];
    __shared__ uint32_t sidx[CL_THREADS];
    const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
    const int s = blockIdx.x / CL_BLOCKS_PER_LIST, part = blockIdx.x % CL_BLOCKS_PER_LIST;
    const int lo = sstart[s];
    int hi = send[s] < list_cap ? send[s] : list_cap;
    const uint32_t want = prefix[s];
    if (want != 0xffffffffu && (long long)lo + (long long)want < (long long)hi) hi = lo + (int)want;
    uint32_t mine = cl_colour_span(f, lists, lo + part * CL_THREADS + wbase, hi, CL_BLOCKS_PER_LIST * CL_THREADS, colrow, rec, srow, sidx);
    // (one counter per list: atomics on ONE address serialise at ~12 ns each -- 8640 waves would cost 0.1 ms)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d, 64);
    if (lane == 0 && mine) atomicAdd(&evals[s], mine);
}

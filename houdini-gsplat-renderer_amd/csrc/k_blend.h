// k_blend.h -- K6: per-tile front-to-back compositing with in-kernel tile filtering.
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
// Data flow (scan -> compact -> composite): the tile walks the depth-ordered list of its
// SUPER-tile 1024 entries at a time (idx + packed tile rect, 8 B each, coalesced, prefetched one
// step ahead); entries whose rect contains this tile are COMPACTED, in list order, into an LDS hit
// queue (wave ballots + a 16-entry cross-wave prefix).  Compositing then runs in rounds of BL_ROUND
// hits: each thread gathers one 48-B record (3 x dwordx4), computes a 4-bit quadrant-overlap mask
// (separating-axis test) and appends the record, still in depth order, to the LDS list of every
// quadrant it reaches (ballot ranks + a cross-wave prefix); each wave then walks only its own
// list, two records per iteration in packed FP32, the pair pre-interleaved in LDS so the operands
// arrive as register pairs.  Scanning is 4 entries/thread/step and SAT/staging run on fully
// populated waves, so sparse lists cost little.
// Early-out: a PIXEL stops accumulating once 1-A < 2^-14 (dropped contribution
// <= 2^-14 * max colour, inside the 1e-3 budget; being per pixel it does not
// depend on chunking, so sharded and unsharded frames are bit-identical); a wave
// stops when all its pixels have, and the workgroup stops reading its list when
// all four waves have.
#pragma once
#include "gsr_device.h"
#include "k_preprocess.h"   // gsr_splat_colour_from_row (on-demand colour of the lazy path)

#ifndef BL_ROUND
#define BL_ROUND 64           // records composited per round.  The kernel wants waves per SIMD: measured on C4
                              // 256 -> 0.40 ms (57 KB LDS, 2 workgroups/CU), 128 -> 0.295, 64 -> 0.278, 32 -> 0.30 (barriers)
#endif
#ifndef BL_WAVES_PER_EU
#define BL_WAVES_PER_EU 6     // 80 VGPRs (8 dwords of spill) instead of 94: 0.278 -> 0.270 ms
#endif
#define BL_SCAN_K 4           // list entries scanned per thread per scan step
#define BL_QCAP 2048          // hit-queue ring capacity (>= BL_ROUND + 2 * BL_SCAN_K * 256)
#define BL_PAIR_F4_MAX 6      // float4s per staged record PAIR: 5 (80 B), 6 with the depth test
#define GSR_T_MIN 6.103515625e-05f  // 2^-14

struct GsrBlendArgs {
    int32_t width, height;      // full image
    int32_t tiles_x;            // tiles per row
    int32_t local_tiles;        // tiles_x * local_tiles_y
    GsrShard shard;             // which tile rows this launch owns
    int32_t band_rows;          // pixel rows of the output (band) image
    int32_t super_shift;        // log2(super-tile edge in tiles)
    int32_t stiles_x;
    int32_t use_map;            // blockIdx -> tile through tile_map (XCD-aware order)
    int32_t flags;              // GSR_FLAG_*
    int32_t list_cap;           // entries the list buffer holds (a speculative launch may see ranges beyond it)
};

// Staging layout: one list PER QUADRANT (= per wave), holding the round's records that reach that
// quadrant, in depth order, two records interleaved per block so that every ds_read_b128 lands as
// ready-made operand pairs of the packed-FP32 instructions (no v_mov shuffling):
//   f4 0: a.a1x b.a1x a.a1y b.a1y     f4 3: a.r a.g a.b a.opacity
//   f4 1: a.b1x b.b1x a.b1y b.b1y     f4 4: b.r b.g b.b b.opacity
//   f4 2: a.c0  b.c0  a.c1  b.c1      f4 5: a.zwin b.zwin - -          (HAS_DEPTH only)
// c0/c1 = the two affine forms of the record at the TILE origin (contract v2): kq0 = lx*a1x + ly*a1y + c0 for the
// pixel (lx, ly) of the tile -- computed once per (record, tile) by the gathering thread.
// HAS_DEPTH = false compiles the depth compare and the sixth float4 out of the inner loop.
#if BL_WAVES_PER_EU > 0
#define BL_OCC __attribute__((amdgpu_waves_per_eu(BL_WAVES_PER_EU)))
#else
#define BL_OCC
#endif
// Lazy colour (k_colour.h): a record whose colour is still pending makes the plain kernel (LAZY = false) give the tile up
// -- it is appended to `redo` -- and the LAZY = true instantiation, launched right behind, composites exactly those tiles
// with the colour evaluated on demand in the gather.
struct GsrLazyArgs {
    GsrFrame f;
    const uint4* colrow;       // 128-byte row per splat: position + colour halves
    int32_t* redo;             // tiles given up by the plain kernel
    uint32_t* redo_count;
};
template <bool HAS_DEPTH, bool LAZY>
__device__ __forceinline__ void
gsr_blend_tile(const GsrBlendArgs& a, const int32_t* __restrict__ tile_map, const uint2* __restrict__ svals,
               const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
               const GsrRecord* __restrict__ recs, float4* __restrict__ out, uint4* __restrict__ tile_work,
               const float* __restrict__ zwin, const float* __restrict__ depth, const GsrLazyArgs& lz)
{
    constexpr int PF4 = HAS_DEPTH ? 6 : 5;
    __shared__ float4 slist[4][(BL_ROUND / 2) * PF4];
    __shared__ uint32_t q[BL_QCAP];   // hit queue: splat indices in list (= depth) order
    __shared__ uint32_t scnt[2][BL_SCAN_K][4];
    __shared__ unsigned long long swcnt[2][4];   // per gathering wave: 4 x 16-bit counts of records reaching quadrant 0..3
    __shared__ uint32_t sdone[2][4];
    __shared__ uint32_t sfetched, sevals, sredo;
    __shared__ float4 spark[BL_ROUND];   // the gathered (r, g, b, opacity) waits here between gather and staging: LDS
                                         // instead of three VGPRs the register allocator would spill to scratch

    // Workgroup b runs on XCD b%8 (observed dispatch order, MI355X guide): tile_map hands each
    // XCD whole super-tiles, whose 64 tiles read the same list and gather the same records.
    int tile;
    if (LAZY) {
        if (blockIdx.x >= *lz.redo_count) return;
        tile = lz.redo[blockIdx.x];
    } else {
        tile = a.use_map ? tile_map[blockIdx.x] : (int)blockIdx.x;
    }
    if (tile < 0 || tile >= a.local_tiles) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tile % a.tiles_x, lty = tile / a.tiles_x;
    const int gty = gsr_shard_global_row(a.shard, lty);
    const int px = tx * GSR_TILE_PX + (wave & 1) * 8 + (lane & 7);
    const int py = gty * GSR_TILE_PX + (wave >> 1) * 8 + (lane >> 3);
    const bool pix_ok = (px < a.width) && (py < a.height);
    // pixel index inside the tile (contract v2: fragment positions are relative to the tile origin)
    const gsr_v2f lx = (gsr_v2f)((float)((wave & 1) * 8 + (lane & 7))), ly = (gsr_v2f)((float)((wave >> 1) * 8 + (lane >> 3)));
    // tile bounds in pixel-centre coordinates, for the quadrant masks
    const float tcx0 = (float)(tx * GSR_TILE_PX) + 0.5f, tcy0 = (float)(gty * GSR_TILE_PX) + 0.5f;
    if (tid == 0) { sfetched = 0; sevals = 0; sredo = 0; }
    uint32_t my_evals = 0;            // (wave-uniform) records this wave evaluated for its 64 pixels
    // depth test against what the opaque pass left (depth writes stay off): a fragment survives iff its quad's
    // window depth <= depth[pixel] (src/GSplatRenderer.C:595-610).  No depth buffer = +inf.
    const float dpx = (HAS_DEPTH && pix_ok) ? depth[(size_t)py * a.width + px] : __builtin_inff();

    const int st = (gty >> a.super_shift) * a.stiles_x + (tx >> a.super_shift);
    const int s = sstart[st];
    const int e_ = send[st] < a.list_cap ? send[st] : a.list_cap;
    const int n = e_ > s ? e_ - s : 0;

    gsr_v2f C01 = {0.0f, 0.0f};   // {C0, C1} as a register pair
    float C2 = 0.0f, T = 1.0f;    // blue, transmittance 1 - A
    bool wave_done = false;
    uint32_t my_fetched = 0;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    auto tile_in = [&](uint32_t rc) {
        const int x0 = rc & 255, y0 = (rc >> 8) & 255, x1 = (rc >> 16) & 255, y1 = rc >> 24;
        return tx >= x0 && tx <= x1 && gty >= y0 && gty <= y1;
    };

    // ---- scan state (identical in every thread)
    int scan_pos = 0;                 // next list entry to scan
    uint32_t q_head = 0, q_tail = 0;  // monotonic; slot = counter & (BL_QCAP-1)
    int spar = 0;
    uint2 pre[BL_SCAN_K];             // entries of the next scan step, prefetched
#pragma unroll
    for (int k = 0; k < BL_SCAN_K; ++k) {
        const int i = k * 256 + tid;
        pre[k] = (i < n) ? svals[s + i] : make_uint2(0u, GSR_RECT_EMPTY);
    }
    int round = 0;
    bool saturated = false;           // left because every pixel is opaque, not because the list ended

    for (;;) {
        // (1) SCAN until a round's worth of hits is queued, or the list ends
        bool scanned_any = false;
        while ((int)(q_tail - q_head) < BL_ROUND && scan_pos < n) {
            uint32_t rank[BL_SCAN_K];
            bool hit[BL_SCAN_K];
#pragma unroll
            for (int k = 0; k < BL_SCAN_K; ++k) {
                hit[k] = tile_in(pre[k].y);          // padding entries carry an empty rect
                const unsigned long long bal = __ballot(hit[k]);
                rank[k] = (uint32_t)__builtin_popcountll(bal & lt_mask);
                if (lane == 0) scnt[spar][k][wave] = (uint32_t)__builtin_popcountll(bal);
            }
            __syncthreads();
            uint32_t tot = 0, mybase[BL_SCAN_K];
#pragma unroll
            for (int k = 0; k < BL_SCAN_K; ++k)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (w == wave) mybase[k] = tot;
                    tot += scnt[spar][k][w];
                }
#pragma unroll
            for (int k = 0; k < BL_SCAN_K; ++k)
                if (hit[k]) q[(q_tail + mybase[k] + rank[k]) & (BL_QCAP - 1)] = pre[k].x;
            q_tail += tot;
            scan_pos += BL_SCAN_K * 256;
            spar ^= 1;
            scanned_any = true;
#pragma unroll
            for (int k = 0; k < BL_SCAN_K; ++k) {   // prefetch the next step's entries
                const int i = scan_pos + k * 256 + tid;
                pre[k] = (i < n) ? svals[s + i] : make_uint2(0u, GSR_RECT_EMPTY);
            }
        }
        if (scanned_any) __syncthreads();            // queue writes visible to every wave

        // (2) gather this round's records: thread t takes the t-th queued hit
        const int avail = (int)(q_tail - q_head);
        const int take = avail < BL_ROUND ? avail : BL_ROUND;
        if (take == 0) break;                         // list exhausted and queue empty
        const bool have = tid < take;
        float4 r0, r1, r2;
        float rz = 0.0f, c0 = 0.0f, c1 = 0.0f;
        uint32_t m = 0;
        if (have) {
            const uint32_t ridx = q[(q_head + tid) & (BL_QCAP - 1)];
            const float4* p = reinterpret_cast<const float4*>(recs + ridx);
            r0 = p[0]; r1 = p[1]; r2 = p[2];
            const uint32_t tag = __builtin_bit_cast(uint32_t, r2.x);
            if (tag == GSR_COLOUR_PENDING) {
                if (LAZY) gsr_splat_colour_from_row(lz.f, lz.colrow, ridx, r2.x, r2.y, r2.z);
                else sredo = 1u;   // the colour pass did not reach this record: give the tile to the LAZY instantiation
            }
            spark[tid] = r2;
            rz = HAS_DEPTH ? zwin[ridx] : 0.0f;
            ++my_fetched;
            // Which 8x8 quadrants can the splat touch?  Separating-axis test of the oriented quad
            // (shrunk to the radius where alpha can still reach 1/255) against each quadrant's box of
            // pixel centres: the box axes (= bbox test) and the quad's own two axes.  Conservative.
            // r0 = (cx, cy, hx, hy), r1 = (a1x, a1y, b1x, b1y) = kappa e/s1, kappa e_perp/s2, r2 = (r, g, b, opacity)
            const float rqk = (((a.flags & GSR_FLAG_NO_ALPHA_RADIUS) ? 2.0f : gsr_support_radius(r2.w)) + 1.0e-3f) * GSR_KAPPA;
            const float bx0 = r0.x - r0.z, bx1 = r0.x + r0.z, by0 = r0.y - r0.w, by1 = r0.y + r0.w;
            const bool xl = bx0 <= tcx0 + 7.0f, xr = bx1 >= tcx0 + 8.0f;
            const bool yb = by0 <= tcy0 + 7.0f, yt = by1 >= tcy0 + 8.0f;
            // radius of a quadrant's box of pixel centres along each (scaled) quad axis
            const float lim1 = rqk + 3.5f * (__builtin_fabsf(r1.x) + __builtin_fabsf(r1.y));
            const float lim2 = rqk + 3.5f * (__builtin_fabsf(r1.z) + __builtin_fabsf(r1.w));
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float ddx = (tcx0 + 3.5f + 8.0f * (float)(qd & 1)) - r0.x;
                const float ddy = (tcy0 + 3.5f + 8.0f * (float)(qd >> 1)) - r0.y;
                const float pu = __builtin_fabsf(ddx * r1.x + ddy * r1.y);
                const float pv = __builtin_fabsf(ddx * r1.z + ddy * r1.w);
                const bool box = ((qd & 1) ? xr : xl) && ((qd >> 1) ? yt : yb);
                if (box && (((a.flags & GSR_FLAG_NO_SAT) != 0) || (pu <= lim1 && pv <= lim2))) m |= 1u << qd;
            }
            // the record's two affine forms at the tile origin (contract v2, same operations as the oracle)
            const float d0x = tcx0 - r0.x, d0y = tcy0 - r0.y;
            c0 = gsr_fma(d0x, r1.x, d0y * r1.y);
            c1 = gsr_fma(d0x, r1.z, d0y * r1.w);
        }
        // (3) per-quadrant list positions: rank inside this gathering wave now, wave bases after the barrier
        uint32_t rnk[4];
        unsigned long long wcnt = 0;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const unsigned long long bal = __ballot((m >> qd) & 1u);
            rnk[qd] = (uint32_t)__builtin_popcountll(bal & lt_mask);
            wcnt |= (unsigned long long)__builtin_popcountll(bal) << (16 * qd);
        }
        const int rpar = round & 1;
        if (lane == 0) { swcnt[rpar][wave] = wcnt; sdone[rpar][wave] = wave_done ? 1u : 0u; }
        __syncthreads();
        if (!LAZY && sredo) {   // (uniform: read after the barrier that follows every gather)
            if (tid == 0 && lz.redo) lz.redo[atomicAdd(lz.redo_count, 1u)] = tile;
            return;
        }
        const bool block_done = (sdone[rpar][0] & sdone[rpar][1] & sdone[rpar][2] & sdone[rpar][3]) != 0u;
        if (block_done) { saturated = true; break; }
        unsigned long long base = 0, total = 0;   // 4 x 16-bit fields (a field is at most BL_ROUND)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const unsigned long long c = swcnt[rpar][g];
            if (g < wave) base += c;
            total += c;
        }
        if (m) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                if (!((m >> qd) & 1u)) continue;
                const uint32_t pos = (uint32_t)((base >> (16 * qd)) & 0xffffu) + rnk[qd];
                float* blk = reinterpret_cast<float*>(&slist[qd][(pos >> 1) * PF4]);
                const uint32_t h = pos & 1u;
                blk[0 + h] = r1.x; blk[2 + h] = r1.y; blk[4 + h] = r1.z; blk[6 + h] = r1.w;
                blk[8 + h] = c0; blk[10 + h] = c1;
                reinterpret_cast<float4*>(blk)[3 + h] = spark[tid];
                if (HAS_DEPTH) blk[20 + h] = rz;
            }
        }
        if (tid < 4) {   // odd list: pad with a record that cannot contribute (opacity 0 -> alpha 0 < 1/255)
            const uint32_t cnt = (uint32_t)((total >> (16 * tid)) & 0xffffu);
            if (cnt & 1u) {
                float* blk = reinterpret_cast<float*>(&slist[tid][(cnt >> 1) * PF4]);
                blk[1] = 0.0f; blk[3] = 0.0f; blk[5] = 0.0f; blk[7] = 0.0f;
                blk[9] = 0.0f; blk[11] = 0.0f;
                // colour and opacity 0 (the axes may be stale garbage: a NaN there is rejected by the quad test).  The zero is
                // made in place: hipcc would otherwise keep a float4 of zeros live across the whole loop -- and spill it.
                float z;
                asm volatile("v_mov_b32 %0, 0" : "=v"(z));
                blk[16] = z; blk[17] = z; blk[18] = z; blk[19] = z;
                if (HAS_DEPTH) blk[21] = 0.0f;
            }
        }
        q_head += (uint32_t)take;
        __syncthreads();

        // (4) composite: wave w walks ITS list, two records per iteration in packed FP32
        if (!wave_done) {
            // (wave-uniform by construction; readfirstlane tells the compiler, so the loop runs on the scalar unit)
            const int cnt = __builtin_amdgcn_readfirstlane((int)((total >> (16 * wave)) & 0xffffu));
            const int npairs = (cnt + 1) >> 1;
            const float4* L = slist[wave];
            int p = 0;
            while (p < npairs) {
                const int pend = (p + 32 < npairs) ? p + 32 : npairs;
                for (; p < pend; ++p) {
                    const float4 v0 = L[p * PF4 + 0], v1 = L[p * PF4 + 1], v2 = L[p * PF4 + 2];
                    const float4 v3 = L[p * PF4 + 3], v4 = L[p * PF4 + 4];
                    // kappa * (quad-local coordinate) of this pixel for the two records
                    const gsr_v2f q0 = gsr_fma2(lx, (gsr_v2f){v0.x, v0.y}, gsr_fma2(ly, (gsr_v2f){v0.z, v0.w}, (gsr_v2f){v2.x, v2.y}));
                    const gsr_v2f q1 = gsr_fma2(lx, (gsr_v2f){v1.x, v1.y}, gsr_fma2(ly, (gsr_v2f){v1.z, v1.w}, (gsr_v2f){v2.z, v2.w}));
                    const gsr_v2f pw = gsr_fma2(q0, q0, q1 * q1);
                    // (lanes outside the quad may feed exp2 a large negative argument: their result is unused)
                    const gsr_v2f e = gsr_exp2n2(-pw);
                    // clamp(exp * opacity, 0, 1): folds into the multiply's clamp modifier
                    const float ala = __builtin_fminf(__builtin_fmaxf(e.x * v3.w, 0.0f), 1.0f);
                    const float alb = __builtin_fminf(__builtin_fmaxf(e.y * v4.w, 0.0f), 1.0f);
                    bool ina = (__builtin_fmaxf(__builtin_fabsf(q0.x), __builtin_fabsf(q1.x)) <= GSR_QLIM) && (ala >= (1.0f / 255.0f));
                    bool inb = (__builtin_fmaxf(__builtin_fabsf(q0.y), __builtin_fabsf(q1.y)) <= GSR_QLIM) && (alb >= (1.0f / 255.0f));
                    if (HAS_DEPTH) {
                        const float4 v5 = L[p * PF4 + 5];
                        ina = ina && (v5.x <= dpx);
                        inb = inb && (v5.y <= dpx);
                    }
                    // branch-free under-blend: a rejected fragment blends weight 0, which leaves C and T
                    // bit-identical and costs no exec-mask juggling on the scalar unit.  w = (1-A)*alpha once;
                    // {C0,C1} update as a register pair.
                    // a PIXEL stops accumulating once T < 2^-14: per pixel, so the image is a pure function of the
                    // depth-ordered records -- independent of rounds, quadrant masks, super-tiles and shards
                    const float wa = T * ((ina && T >= GSR_T_MIN) ? ala : 0.0f);
                    C01 = gsr_fma2((gsr_v2f)(wa), (gsr_v2f){v3.x, v3.y}, C01);
                    C2 = gsr_fma(wa, v3.z, C2);
                    T = T - wa;
                    const float wb = T * ((inb && T >= GSR_T_MIN) ? alb : 0.0f);
                    C01 = gsr_fma2((gsr_v2f)(wb), (gsr_v2f){v4.x, v4.y}, C01);
                    C2 = gsr_fma(wb, v4.z, C2);
                    T = T - wb;
                }
                if (__all(!pix_ok || T < GSR_T_MIN)) { wave_done = true; break; }
            }
            const int evald = 2 * p;
            my_evals += (uint32_t)(evald < cnt ? evald : cnt);
        }
        ++round;
        __syncthreads();   // the lists (and consumed queue slots) may be overwritten from here on
    }
    if (pix_ok) {
        // (pixel coordinates re-derived from an opaque copy of the thread id: keeping them live across the loop costs a spill)
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int w2 = t2 >> 6, l2 = t2 & 63;
        const int brow = lty * GSR_TILE_PX + (w2 >> 1) * 8 + (l2 >> 3);
        const int bcol = tx * GSR_TILE_PX + (w2 & 1) * 8 + (l2 & 7);
        out[(size_t)brow * a.width + bcol] = make_float4(C01.x, C01.y, C2, 1.0f - T);
    }
    // bookkeeping for the roofline: list entries scanned and records gathered by this tile
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) my_fetched += __shfl_down(my_fetched, d, 64);
    __syncthreads();  // orders the sfetched = 0 store when the list was empty
    if (lane == 0) { atomicAdd(&sfetched, my_fetched); atomicAdd(&sevals, my_evals); }
    __syncthreads();
    if (tid == 0) {
        // entries actually read: everything up to scan_pos plus the prefetched step
        const int rd = scan_pos + BL_SCAN_K * 256;
        tile_work[tile] = make_uint4((uint32_t)(rd < n ? rd : n), sfetched, sevals, saturated ? 1u : 0u);
    }
}

// the two entry points: the plain kernel at 6 waves per SIMD; the fallback carries the SH evaluation and is left to the
// register allocator (it runs for the few tiles the colour pass did not cover)
template <bool HAS_DEPTH>
__global__ void __launch_bounds__(256) BL_OCC
k_blend(GsrBlendArgs a, const int32_t* __restrict__ tile_map, const uint2* __restrict__ svals,
        const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
        const GsrRecord* __restrict__ recs, float4* __restrict__ out, uint4* __restrict__ tile_work,
        const float* __restrict__ zwin, const float* __restrict__ depth, GsrLazyArgs lz)
{
    gsr_blend_tile<HAS_DEPTH, false>(a, tile_map, svals, sstart, send, recs, out, tile_work, zwin, depth, lz);
}
template <bool HAS_DEPTH>
__global__ void __launch_bounds__(256)
k_blend_lazy(GsrBlendArgs a, const int32_t* __restrict__ tile_map, const uint2* __restrict__ svals,
             const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
             const GsrRecord* __restrict__ recs, float4* __restrict__ out, uint4* __restrict__ tile_work,
             const float* __restrict__ zwin, const float* __restrict__ depth, GsrLazyArgs lz)
{
    gsr_blend_tile<HAS_DEPTH, true>(a, tile_map, svals, sstart, send, recs, out, tile_work, zwin, depth, lz);
}

// One workgroup sums the per-tile bookkeeping into counters[1] (records gathered, this frame),
// [2] (records, running total), [3] (entries scanned, this frame), [4] (entries, running total),
// [5] (wave-record evaluations, running total) -- a handful of atomics per frame instead of per
// tile (same-address atomics serialise at ~12 ns each on MI355X).  Last kernel of a frame.
#define SW_THREADS 1024
#define SW_UNROLL 8
#ifndef SW_HEADROOM_SHIFT
#define SW_HEADROOM_SHIFT 2
#endif
#ifndef SW_HEADROOM_ADD
#define SW_HEADROOM_ADD 1024u
#endif
#define SW_HEADROOM_SHIFT_ SW_HEADROOM_SHIFT
#define SW_HEADROOM_ADD_ SW_HEADROOM_ADD
struct GsrSumArgs {
    int32_t n_tiles, tiles_x, super_shift, stiles_x, n_super;
    GsrShard shard;
};
__global__ void __launch_bounds__(SW_THREADS)
k_sum_work(const uint4* __restrict__ tile_work, GsrSumArgs g, unsigned long long* __restrict__ counters,
           const uint32_t* __restrict__ n_visible, unsigned long long* __restrict__ summary /* device [8]: fetched by gsr_get_stats */,
           uint32_t* __restrict__ prefix /* [256] lazy colour: list entries to colour per super-tile, next frame (or NULL) */,
           const uint32_t* __restrict__ redo_count /* tiles the plain blend kernel gave up this frame (or NULL) */,
           uint32_t* __restrict__ colour_evals /* [256] per-list counts of the colour pass, cleared here (or NULL) */,
           unsigned long long* __restrict__ colour_total /* running total of the above */,
           const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
           uint32_t* __restrict__ lazy_hint /* would lazy colour pay for a frame like this one? (read by the next frames) */)
{
    __shared__ unsigned long long s_sum[3];
    __shared__ uint32_t s_max[256];
    __shared__ uint32_t s_unsat, s_est, s_cev;
    // Everything the tail of this kernel needs from memory is fetched NOW, next to the tile_work loads: the kernel is one
    // workgroup at the very end of the frame, and every dependent global round trip in it (~1-2 us) is frame latency.
    unsigned long long old2 = 0, old4 = 0, old5 = 0, old_ct = 0;
    uint32_t nvis = 0, nredo = 0, my_len = 0, my_cev = 0;
    if (threadIdx.x == 0) {
        old2 = counters[2]; old4 = counters[4]; old5 = counters[5];
        nvis = *n_visible;
        nredo = redo_count ? *redo_count : 0u;
        if (colour_evals) old_ct = *colour_total;
    }
    if (prefix && (int)threadIdx.x < g.n_super) my_len = (uint32_t)(send[threadIdx.x] - sstart[threadIdx.x]);
    if (colour_evals && threadIdx.x < 256) my_cev = colour_evals[threadIdx.x];
    if (threadIdx.x == 0) { s_unsat = 0; s_est = 0; s_cev = 0; }
    if (threadIdx.x < 3) s_sum[threadIdx.x] = 0;
    if (threadIdx.x < 256) s_max[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long sc = 0, fe = 0, ev = 0;
    uint32_t unsat = 0;   // tiles that composited something and ran to the end of their list: lazy colour sends them to the fallback
    for (int i0 = 0; i0 < g.n_tiles; i0 += SW_UNROLL * SW_THREADS) {
        uint4 w[SW_UNROLL];
#pragma unroll
        for (int u = 0; u < SW_UNROLL; ++u) {   // independent loads: one memory round trip per 8192 tiles
            const int i = i0 + u * SW_THREADS + (int)threadIdx.x;
            w[u] = i < g.n_tiles ? tile_work[i] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < SW_UNROLL; ++u) {
            sc += w[u].x; fe += w[u].y; ev += w[u].z;
            const int i = i0 + u * SW_THREADS + (int)threadIdx.x;
            // deepest scan among the tiles of each super-tile that SATURATED: a tile that ran to the end of its list (the
            // cloud's silhouette) would ask for the whole list; such tiles have few hits and take the on-demand fallback
            if (i < g.n_tiles && !w[u].w && w[u].y) ++unsat;
            if (prefix && i < g.n_tiles && w[u].w) {
                const int tx = i % g.tiles_x, gty = gsr_shard_global_row(g.shard, i / g.tiles_x);
                atomicMax(&s_max[(gty >> g.super_shift) * g.stiles_x + (tx >> g.super_shift)], w[u].x);
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        sc += __shfl_down(sc, d, 64); fe += __shfl_down(fe, d, 64); ev += __shfl_down(ev, d, 64); unsat += __shfl_down(unsat, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_sum[0], sc); atomicAdd(&s_sum[1], fe); atomicAdd(&s_sum[2], ev); atomicAdd(&s_unsat, unsat); }
    __syncthreads();
    if (prefix && (int)threadIdx.x < g.n_super) {
        const uint32_t m = s_max[threadIdx.x];
        {   // how many colour evaluations the lazy pass would make for a frame like this one
            const uint32_t want = m + (m >> SW_HEADROOM_SHIFT_) + SW_HEADROOM_ADD_;
            atomicAdd(&s_est, want < my_len ? want : my_len);
        }
        prefix[threadIdx.x] = m + (m >> SW_HEADROOM_SHIFT) + SW_HEADROOM_ADD;   // headroom for the next frame's camera move
    }
    if (colour_evals && threadIdx.x < 256) {
        colour_evals[threadIdx.x] = 0u;
        if (my_cev) atomicAdd(&s_cev, my_cev);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // Lazy colour pays when the colour pass would evaluate well under half of what eager evaluation does (it gathers
        // rows at random, eager streams them) and (almost) no tile would need the on-demand fallback.  Small or sparse
        // clouds (BASELINE C2, C3) fail one of the two; the 6 M-splat scenes pass both.
        if (lazy_hint) *lazy_hint = (prefix && (unsigned long long)s_est * 10ull < (unsigned long long)nvis * 4ull &&
                                     s_unsat * 64u <= (uint32_t)g.n_tiles) ? 1u : 0u;
        // running totals: plain read-modify-write (a slot's frames are serialised on its stream; nothing else touches them)
        const unsigned long long t2 = old2 + s_sum[1], t4 = old4 + s_sum[0], t5 = old5 + s_sum[2];
        counters[1] = s_sum[1]; counters[2] = t2; counters[3] = s_sum[0]; counters[4] = t4; counters[5] = t5;
        if (colour_evals) *colour_total = old_ct + s_cev;
        // the frame's summary stays in device memory (gsr_get_stats copies 64 bytes after its stream sync): writing it
        // to mapped host memory every frame cost ~10 us of PCIe round trips at the end of the frame
        summary[0] = 0ull; summary[1] = s_sum[1]; summary[2] = t2; summary[3] = s_sum[0]; summary[4] = t4; summary[5] = t5;
        summary[6] = (unsigned long long)nvis;
        summary[7] = (unsigned long long)nredo;
    }
}

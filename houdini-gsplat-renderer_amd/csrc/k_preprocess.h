// k_preprocess.h -- K0 (repack at upload) and K1 (per-splat vertex stage).
//
// K1 replaces the reference's main vertex shader
// (/root/reference/gsplat_plugin/shaders/GSplatShaderSource.h:190-288 with the
// helpers in shaders/GSplatShaderCoreLib.h:10-93,103-179) -- evaluated ONCE per
// splat instead of once per quad corner -- and the CPU distance loop of
// argsortByDistance (src/GSplatRenderer.C:194-204).  One wavefront per cluster that survived k_cluster_cull (k_cluster.h):
// 32 B read per splat of it, 56 B (record, key, payload) written per splat that stays, + its 96 B of colour halves when the
// frame shades in K1 (occlusion-culled frames; unculled ones leave the colours to k_colour.h).  Bound by FP32 issue -- the
// ~1000-instruction covariance chain with its eleven IEEE divisions and four square roots -- not by HBM.
#pragma once
#include "gsr_device.h"
#include "k_cluster.h"

#ifndef GSR_K1_THREADS
#define GSR_K1_THREADS 256
#endif

// ---------------------------------------------------------------------------
// K0 (round 6): the registerUpdate()-layout arrays of a whole upload (in the staging arena, device memory, upload order) -> the
// resident layout, straight INTO STORAGE ORDER, with the cluster bounds on the way out.  Runs once per geometry change.
//   storage slot j <- splat perm[j] (the stable order of the positions' Morton codes, k_cluster.h; NULL: upload order)
// One workgroup of 512 threads = 64 slots = ONE CLUSTER, eight lanes per splat; lane q of a splat's eight:
//   q = 0      P (12 B) + opacity (4 B)            -> geoA[j], and colrow[j][0]
//   q = 1..6   16 bytes of the x / y / z SH rows   -> LDS; then chunk c = q - 1 of the 48 colour halves (Cd.rgb, sh1.rgb, ..., sh15.rgb:
//              half h = 3 * coefficient + channel) <- LDS: a 16 x 3 -> 3 x 16 transpose per splat -> col[c][j] and colrow[j][1 + c]
//   q = 7      scale (6 B) + orient (8 B)          -> geoB[j] with its extent bound; colrow[j][7] = 0
// so a wave WRITES eight whole 128-byte colrow lines, 128 contiguous bytes of geoA, of geoB and of each colour chunk, and READS, per
// splat, its rows of the source arrays (96 + 36 bytes; a gather when perm is a permutation).  Before round 6 this was three passes:
// k_repack (one thread per splat, 2-byte loads at a 32-byte stride, 16-byte stores at a 128-byte stride: 1.5 x write amplification,
// 0.19 of HBM), then k_permute_geo + k_permute_rows over a second copy of the geometry, then k_cluster_bounds.
struct GsrPackSrc {
    const float* P; const float* alpha;
    const uint16_t *Cd, *scale, *orient, *shx, *shy, *shz;
};
#define GSR_PACK_THREADS 512
template <bool SH>
__global__ void __launch_bounds__(GSR_PACK_THREADS)
k_pack(uint32_t n, uint32_t cap, GsrPackSrc src, const uint32_t* __restrict__ perm,
       float4* __restrict__ geoA, uint4* __restrict__ geoB, uint4* __restrict__ col, uint4* __restrict__ colrow,
       float4* __restrict__ clusA, float4* __restrict__ clusB)
{
    static_assert(GSR_PACK_THREADS == 8 * GSR_CLUSTER, "one workgroup = one cluster");
    __shared__ __attribute__((aligned(16))) uint16_t s_h[GSR_CLUSTER][56];   // per splat: x[16] y[16] z[16] Cd[3] (+ pad: 112 B rows)
    __shared__ float s_red[8][8];                                           // per wave: lo.xyz hi.xyz mf bad
    const int tid = threadIdx.x, q = tid & 7, sp = tid >> 3, wave = tid >> 6;
    const uint32_t j = blockIdx.x * (uint32_t)GSR_CLUSTER + (uint32_t)sp;
    const bool live = j < n;
    const uint32_t i = live ? (perm ? perm[j] : j) : 0u;
    auto pk = [](uint16_t lo, uint16_t hi) { return (uint32_t)lo | ((uint32_t)hi << 16); };
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, mf = 0.0f;
    bool bad = false;
    uint4 rowpiece = make_uint4(0u, 0u, 0u, 0u);       // this lane's eighth of the splat's 128-byte row (lane 7: padding)
    if (live) {
        if (q == 0) {
            const float px = src.P[3 * (size_t)i], py = src.P[3 * (size_t)i + 1], pz = src.P[3 * (size_t)i + 2], op = src.alpha[i];
            geoA[j] = make_float4(px, py, pz, op);
            rowpiece = make_uint4(__float_as_uint(px), __float_as_uint(py), __float_as_uint(pz), __float_as_uint(op));
            s_h[sp][48] = src.Cd[3 * (size_t)i]; s_h[sp][49] = src.Cd[3 * (size_t)i + 1]; s_h[sp][50] = src.Cd[3 * (size_t)i + 2];
            bad = !(__builtin_fabsf(px) < 3.0e38f) || !(__builtin_fabsf(py) < 3.0e38f) || !(__builtin_fabsf(pz) < 3.0e38f);
            lo[0] = hi[0] = px; lo[1] = hi[1] = py; lo[2] = hi[2] = pz;
        } else if (q == 7) {
            const uint16_t* s = src.scale + 3 * (size_t)i;
            const uint16_t* o = src.orient + 4 * (size_t)i;
            const uint16_t s0 = s[0], s1 = s[1], s2 = s[2], o0 = o[0], o1 = o[1], o2 = o[2], o3 = o[3];
            // eighth half of geoB: an upper bound of |diag(scale) R(orient)^T|_F, the only thing K1's cheap extent bound needs of the
            // scale and the (unnormalised) quaternion -- so the bound costs a dozen instructions per splat instead of ninety
            uint16_t mf_h;
            {
                const float sx = gsr_h2f(s0), sy = gsr_h2f(s1), sz = gsr_h2f(s2);
                const float qi = gsr_h2f(o0), qj = gsr_h2f(o1), qk = gsr_h2f(o2), qr = gsr_h2f(o3);
                const float r00 = 1.0f - 2.0f * gsr_fma(qj, qj, qk * qk), r01 = 2.0f * gsr_fma(qi, qj, -(qr * qk)), r02 = 2.0f * gsr_fma(qi, qk, qr * qj);
                const float r10 = 2.0f * gsr_fma(qi, qj, qr * qk), r11 = 1.0f - 2.0f * gsr_fma(qi, qi, qk * qk), r12 = 2.0f * gsr_fma(qj, qk, -(qr * qi));
                const float r20 = 2.0f * gsr_fma(qi, qk, -(qr * qj)), r21 = 2.0f * gsr_fma(qj, qk, qr * qi), r22 = 1.0f - 2.0f * gsr_fma(qi, qi, qj * qj);
                const float mf2 = sx * sx * (r00 * r00 + r10 * r10 + r20 * r20) + sy * sy * (r01 * r01 + r11 * r11 + r21 * r21) +
                                  sz * sz * (r02 * r02 + r12 * r12 + r22 * r22);
                const float mfv = __builtin_sqrtf(mf2) * 1.001f;
                const _Float16 hh = (_Float16)mfv;              // rounded up (to nearest, then one step if that fell short): mf >= 0
                mf_h = __builtin_bit_cast(uint16_t, hh);
                if ((float)hh < mfv) mf_h += 1;                 // (0x7bff + 1 = inf; inf / NaN stay what they are: K1 then takes the full path)
            }
            geoB[j] = make_uint4(pk(s0, s1), pk(s2, o0), pk(o1, o2), pk(o3, mf_h));
            mf = gsr_h2f(mf_h);
            bad = !(mf < 6.0e4f);
        } else if (SH) {
            // lanes 1..6: x[0..7] x[8..15] y[0..7] y[8..15] z[0..7] z[8..15] (rows of 16 halves = 32 bytes, 16-byte loads)
            const int ch = (q - 1) >> 1, part = (q - 1) & 1;
            const uint16_t* row = (ch == 0 ? src.shx : (ch == 1 ? src.shy : src.shz)) + 16 * (size_t)i + 8 * part;
            *reinterpret_cast<uint4*>(&s_h[sp][16 * ch + 8 * part]) = *reinterpret_cast<const uint4*>(row);
        }
    }
    // (a splat's eight lanes sit in one wave: its LDS row is written and read by that wave only)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (live && q >= 1 && (SH ? q <= 6 : q == 1)) {
        const int c = q - 1;
        uint16_t h[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int hh = 8 * c + k, co = hh / 3, chn = hh - 3 * co;     // coefficient (0 = Cd), channel
            // coefficient co of the reference's row-major 4x4 sits at flat index co - 1 of the x / y / z rows (co = 1..15)
            h[k] = co == 0 ? s_h[sp][48 + chn] : (SH ? s_h[sp][16 * chn + (co - 1)] : (uint16_t)0);
        }
        const uint4 v = make_uint4(pk(h[0], h[1]), pk(h[2], h[3]), pk(h[4], h[5]), pk(h[6], h[7]));
        col[(size_t)c * cap + j] = v;                  // SoA chunks: coalesced for a pass over ALL splats (eager colour)
        rowpiece = v;
    }
    // ... and as ONE 128-byte row per splat, gathered by index (lazy colour): piece q from lane q, the eight pieces of a row in ONE store
    // instruction (as three -- position, colours, padding -- a quarter of the rows went out as partial lines: PMC 1.16 x the output)
    if (SH && live) colrow[(size_t)j * 8 + q] = rowpiece;
    // the cluster's bounds (k_cluster.h: clusA = lo.xyz of the positions + the largest extent bound, clusB = hi.xyz + "never cull" flag)
    const bool any_bad = __ballot(bad) != 0ull;
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {     // (the position lanes are q = 0, the extent lanes q = 7: strides of eight)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = __builtin_fminf(lo[k], __shfl_xor(lo[k], d, 64));
            hi[k] = __builtin_fmaxf(hi[k], __shfl_xor(hi[k], d, 64));
        }
        mf = __builtin_fmaxf(mf, __shfl_xor(mf, d, 64));
    }
    const int lane = tid & 63;
    if (lane == 0) { s_red[wave][0] = lo[0]; s_red[wave][1] = lo[1]; s_red[wave][2] = lo[2]; s_red[wave][3] = hi[0]; s_red[wave][4] = hi[1]; s_red[wave][5] = hi[2]; s_red[wave][7] = any_bad ? 1.0f : 0.0f; }
    if (lane == 7) s_red[wave][6] = mf;
    __syncthreads();
    if (tid == 0) {
        float l0 = 3.0e38f, l1 = 3.0e38f, l2 = 3.0e38f, h0 = -3.0e38f, h1 = -3.0e38f, h2 = -3.0e38f, m = 0.0f, b = 0.0f;
        for (int w = 0; w < 8; ++w) {
            l0 = __builtin_fminf(l0, s_red[w][0]); l1 = __builtin_fminf(l1, s_red[w][1]); l2 = __builtin_fminf(l2, s_red[w][2]);
            h0 = __builtin_fmaxf(h0, s_red[w][3]); h1 = __builtin_fmaxf(h1, s_red[w][4]); h2 = __builtin_fmaxf(h2, s_red[w][5]);
            m = __builtin_fmaxf(m, s_red[w][6]); b = __builtin_fmaxf(b, s_red[w][7]);
        }
        clusA[blockIdx.x] = make_float4(l0, l1, l2, m);
        clusB[blockIdx.x] = make_float4(h0, h1, h2, b != 0.0f ? 1.0f : 0.0f);
    }
}

// ---------------------------------------------------------------------------
// K0': raw float32 point attributes (as a Houdini detail holds them) -> the half arrays k_pack takes.  What
// GR_PrimGsplat::update does on the CPU in a tbb::parallel_for (/root/reference/gsplat_plugin/src/GR_GSplat.C:302-372):
// fp32 -> fp16 (HDK's fpreal16: round to nearest even, overflow to infinity -- what v_cvt_f16_f32 does), the defaults for
// missing attributes, and the three spherical-harmonics naming schemes -> coefficient j in flat slot j of the x / y / z rows.
__device__ __forceinline__ uint16_t gsr_f2h(float f)
{
    const _Float16 h = (_Float16)f;      // v_cvt_f16_f32: round to nearest even
    return __builtin_bit_cast(uint16_t, h);
}
struct GsrRawSh {
    int32_t scheme;              // 0 none, 1 = one array of `vec3_per_point` vec3 per point, 2 = sh1..sh15 (vec3 arrays), 3 = f_rest_0..44 (float arrays)
    int32_t vec3_per_point;      // scheme 1
    const float* array;          // scheme 1
    const float* ptr[45];        // scheme 2: [0..14] (NULL from the first gap on), scheme 3: [0..44] (likewise)
};
__global__ void __launch_bounds__(256)
k_quantize_raw(uint32_t n, const float* __restrict__ Cd, const float* __restrict__ scale, const float* __restrict__ orient, GsrRawSh sh,
               uint16_t* __restrict__ oCd, uint16_t* __restrict__ oS, uint16_t* __restrict__ oO,
               uint16_t* __restrict__ ox, uint16_t* __restrict__ oy, uint16_t* __restrict__ oz)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        oCd[3 * (size_t)i + k] = Cd ? gsr_f2h(Cd[3 * (size_t)i + k]) : (uint16_t)0;            // missing Cd: black
        oS[3 * (size_t)i + k] = scale ? gsr_f2h(scale[3 * (size_t)i + k]) : (uint16_t)0x3c00;  // missing scale: 1
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) oO[4 * (size_t)i + k] = orient ? gsr_f2h(orient[4 * (size_t)i + k]) : (uint16_t)(k == 3 ? 0x3c00 : 0);   // (0, 0, 0, 1)
    if (sh.scheme == 0) return;
    for (int j = 0; j < 16; ++j) {
        float x = 0.0f, y = 0.0f, z = 0.0f;
        bool have = false;
        if (sh.scheme == 1) {
            have = j < sh.vec3_per_point;
            if (have) { const float* v = sh.array + ((size_t)i * sh.vec3_per_point + j) * 3; x = v[0]; y = v[1]; z = v[2]; }
        } else if (sh.scheme == 2) {
            have = j < 15 && sh.ptr[j];
            if (have) { const float* v = sh.ptr[j] + 3 * (size_t)i; x = v[0]; y = v[1]; z = v[2]; }
        } else if (j < 15) {   // channel-major INRIA layout: (f_rest_j, f_rest_{j+15}, f_rest_{j+30}); a missing array leaves zeros
            have = true;
            x = sh.ptr[j] ? sh.ptr[j][i] : 0.0f; y = sh.ptr[j + 15] ? sh.ptr[j + 15][i] : 0.0f; z = sh.ptr[j + 30] ? sh.ptr[j + 30][i] : 0.0f;
        }
        // (a slot without data is a zero HALF, not the conversion of 0.0f: the same bits, said explicitly)
        ox[16 * (size_t)i + j] = have ? gsr_f2h(x) : (uint16_t)0;
        oy[16 * (size_t)i + j] = have ? gsr_f2h(y) : (uint16_t)0;
        oz[16 * (size_t)i + j] = have ? gsr_f2h(z) : (uint16_t)0;
    }
}

// ---------------------------------------------------------------------------
// SH evaluation for one channel; expressions are written and associated exactly
// as in the oracle (and in shaders/GSplatShaderCoreLib.h:146-175).
__device__ __forceinline__ float gsr_shade_sh(float base, const float* sh, float x, float y, float z, int order)
{
    const float SH_C1 = 0.4886025f;
    const float SH_C2_0 = 1.0925484f, SH_C2_1 = -1.0925484f, SH_C2_2 = 0.3153916f, SH_C2_3 = -1.0925484f,
                SH_C2_4 = 0.5462742f;
    const float SH_C3_0 = -0.5900436f, SH_C3_1 = 2.8906114f, SH_C3_2 = -0.4570458f, SH_C3_3 = 0.3731763f,
                SH_C3_4 = -0.4570458f, SH_C3_5 = 1.4453057f, SH_C3_6 = -0.5900436f;
    float res = base;
    if (order >= 1) {
        res += SH_C1 * (-sh[0] * y + sh[1] * z - sh[2] * x);
        if (order >= 2) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            res += (SH_C2_0 * xy) * sh[3] + (SH_C2_1 * yz) * sh[4] + (SH_C2_2 * (2.0f * zz - xx - yy)) * sh[5] +
                   (SH_C2_3 * xz) * sh[6] + (SH_C2_4 * (xx - yy)) * sh[7];
            if (order >= 3) {
                res += (SH_C3_0 * y * (3.0f * xx - yy)) * sh[8] + (SH_C3_1 * xy * z) * sh[9] +
                       (SH_C3_2 * y * (4.0f * zz - xx - yy)) * sh[10] +
                       (SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * sh[11] +
                       (SH_C3_4 * x * (4.0f * zz - xx - yy)) * sh[12] + (SH_C3_5 * z * (xx - yy)) * sh[13] +
                       (SH_C3_6 * x * (xx - 3.0f * yy)) * sh[14];
            }
        }
    }
    return __builtin_fmaxf(res, 0.0f);
}

__device__ __forceinline__ float aff4(const float* m, float x, float y, float z)
{   // m[0]*x + m[1]*y + m[2]*z + m[3] as the contract's fma chain
    return gsr_fma(m[0], x, gsr_fma(m[1], y, gsr_fma(m[2], z, m[3])));
}
__device__ __forceinline__ float lin3(const float* m, float x, float y, float z)
{
    return gsr_fma(m[0], x, gsr_fma(m[1], y, m[2] * z));
}

// Covariance chain shared by the beauty path (K1) and the wireframe overlay: Sigma = M^T M with
// M = diag(scale) R(q)^T Obj^T, EWA projection with the view matrix (+0.3 low-pass), eigen-decomposition
// -> unit major axis e and the two axis lengths (shaders/GSplatShaderCoreLib.h:10-93).  obm = mat3 of the
// object matrix (rows), or the identity for the wire program, which never applies it.
// Returns false when the projected covariance is not finite (inf/NaN scale or orient halves, or overflow): such a splat
// is dropped -- fminf(NaN, 4096) would otherwise turn it into a screen-filling 4096-px quad.
__device__ __forceinline__ bool gsr_covariance_axes(const GsrFrame& f, const float* obm, float x, float y, float z,
                                                    float sx, float sy, float sz, float qi, float qj, float qk,
                                                    float qr, float& ex, float& ey, float& s1, float& s2)
{
    float R[3][3];
    R[0][0] = 1.0f - 2.0f * gsr_fma(qj, qj, qk * qk);
    R[0][1] = 2.0f * gsr_fma(qi, qj, -(qr * qk));
    R[0][2] = 2.0f * gsr_fma(qi, qk, qr * qj);
    R[1][0] = 2.0f * gsr_fma(qi, qj, qr * qk);
    R[1][1] = 1.0f - 2.0f * gsr_fma(qi, qi, qk * qk);
    R[1][2] = 2.0f * gsr_fma(qj, qk, -(qr * qi));
    R[2][0] = 2.0f * gsr_fma(qi, qk, -(qr * qj));
    R[2][1] = 2.0f * gsr_fma(qj, qk, qr * qi);
    R[2][2] = 1.0f - 2.0f * gsr_fma(qi, qi, qj * qj);
    const float sc[3] = {sx, sy, sz};
    float M0[3][3], Mm[3][3], S[3][3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) M0[p][q] = sc[p] * R[q][p];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            Mm[p][q] = gsr_fma(M0[p][2], obm[q * 3 + 2], gsr_fma(M0[p][1], obm[q * 3 + 1], M0[p][0] * obm[q * 3 + 0]));
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = p; q < 3; ++q) {
            float v = gsr_fma(Mm[2][p], Mm[2][q], gsr_fma(Mm[1][p], Mm[1][q], Mm[0][p] * Mm[0][q]));
            S[p][q] = v;
            S[q][p] = v;
        }

    float tx = aff4(&f.vw[0], x, y, z);
    float ty = aff4(&f.vw[4], x, y, z);
    const float tz = aff4(&f.vw[8], x, y, z);
    {
        float rx = tx / tz, ry = ty / tz;
        rx = __builtin_fminf(__builtin_fmaxf(rx, -f.limx), f.limx);
        ry = __builtin_fminf(__builtin_fmaxf(ry, -f.limy), f.limy);
        tx = rx * tz;
        ty = ry * tz;
    }
    const float j00 = f.focal / tz;
    const float tz2 = tz * tz;
    const float j02 = -(f.focal * tx) / tz2;
    const float j12 = -(f.focal * ty) / tz2;
    float A0[3], A1[3], u0[3], u1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        A0[c] = gsr_fma(j00, f.vw[0 * 4 + c], j02 * f.vw[2 * 4 + c]);
        A1[c] = gsr_fma(j00, f.vw[1 * 4 + c], j12 * f.vw[2 * 4 + c]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        u0[k] = gsr_fma(S[k][2], A0[2], gsr_fma(S[k][1], A0[1], S[k][0] * A0[0]));
        u1[k] = gsr_fma(S[k][2], A1[2], gsr_fma(S[k][1], A1[1], S[k][0] * A1[0]));
    }
    const float cov00 = gsr_fma(A0[2], u0[2], gsr_fma(A0[1], u0[1], A0[0] * u0[0]));
    const float cov01 = gsr_fma(A0[2], u1[2], gsr_fma(A0[1], u1[1], A0[0] * u1[0]));
    const float cov11 = gsr_fma(A1[2], u1[2], gsr_fma(A1[1], u1[1], A1[0] * u1[0]));
    const float ca = cov00 + 0.3f, cb = cov01, cc = cov11 + 0.3f;

    const float mid = 0.5f * (ca + cc);
    const float hd = (ca - cc) * 0.5f;
    const float radius = __builtin_sqrtf(gsr_fma(hd, hd, cb * cb));
    const float lambda1 = mid + radius;
    const float lambda2 = __builtin_fmaxf(mid - radius, 0.1f);
    const float dvx = cb, dvy = lambda1 - ca;
    const float dlen = __builtin_sqrtf(gsr_fma(dvx, dvx, dvy * dvy));
    ex = 1.0f; ey = 0.0f;
    if (dlen > 0.0f) {
        ex = dvx / dlen;
        ey = dvy / dlen;
    }
    s1 = __builtin_fminf(__builtin_sqrtf(2.0f * lambda1), 4096.0f);
    s2 = __builtin_fminf(__builtin_sqrtf(2.0f * lambda2), 4096.0f);
    return __builtin_fabsf(lambda1) < 3.0e38f;   // false for inf and NaN
}

// Colour of one splat: Cd, plus SH evaluated towards the splat when the frame's order is > 0
// (shaders/GSplatShaderSource.h:224,244-274).  cw = the splat's 48 colour halves (Cd.rgb, sh1.rgb ... sh15.rgb) as six
// 16-byte words; (x, y, z) = the splat position after the GSplatOrigin round trip.  One function for the eager path
// (K1), the lazy colour pass and the blend kernel's on-demand fallback: the three produce identical bits.
#define GSR_COLOUR_PENDING 0x7fc0deadu   // bit pattern of a record's `r` while its colour has not been evaluated (a NaN no
                                         // arithmetic produces: fp16->fp32 NaNs have zero low mantissa bits)
__device__ __forceinline__ float gsr_finite_colour(float c)
{
    if (c != c) return 0.0f;
    return c > 3.0e38f ? 3.0e38f : (c < -3.0e38f ? -3.0e38f : c);
}
__device__ __forceinline__ void gsr_splat_colour(const GsrFrame& f, const uint4* cw, float x, float y, float z,
                                                 float& cr, float& cg, float& cbl)
{
    cr = gsr_h2f(cw[0].x & 0xffffu); cg = gsr_h2f(cw[0].x >> 16); cbl = gsr_h2f(cw[0].y & 0xffffu);
    if (f.sh_order > 0) {
        uint32_t w[24];
#pragma unroll
        for (int c = 0; c < 6; ++c) { w[4 * c] = cw[c].x; w[4 * c + 1] = cw[c].y; w[4 * c + 2] = cw[c].z; w[4 * c + 3] = cw[c].w; }
        const float wx = x - f.cam[0], wy = y - f.cam[1], wz = z - f.cam[2];
        const float ox = lin3(&f.io[0], wx, wy, wz);
        const float oy = lin3(&f.io[3], wx, wy, wz);
        const float oz = lin3(&f.io[6], wx, wy, wz);
        const float len = __builtin_sqrtf(gsr_fma(oz, oz, gsr_fma(oy, oy, ox * ox)));
        const float dx = ox / len, dy = oy / len, dz = oz / len;
        // one channel at a time: its fifteen coefficients are decoded, used and dropped before the next channel's (decoded all at once --
        // 45 floats beside the 24 words they come from -- the shading K1 spilt 20 bytes at its 80 registers; the scheduling barriers keep
        // the compiler from interleaving the three evaluations again).  The arithmetic of a channel is untouched.
        float out[3] = {cr, cg, cbl};
#pragma unroll
        for (int chn = 0; chn < 3; ++chn) {
            float sh[15];
#pragma unroll
            for (int j = 0; j < 15; ++j) {
                const int h = 3 * (j + 1) + chn;
                sh[j] = gsr_h2f((w[h >> 1] >> ((h & 1) * 16)) & 0xffffu);
            }
            out[chn] = gsr_shade_sh(out[chn], sh, dx, dy, dz, f.sh_order);
            __builtin_amdgcn_sched_barrier(0);
        }
        cr = out[0]; cg = out[1]; cbl = out[2];
    }
    // (round 6, oracle and kernels together) a colour that is not finite is made finite where it is formed: NaN -> 0, +-inf -> +-3e38.
    // k_blend blends a REJECTED fragment with weight 0 (no exec-mask juggling), and 0 x inf is NaN: one poisoned splat used to turn
    // every pixel of every quadrant it was staged in into NaN -- and whether it was staged depends on culling
    cr = gsr_finite_colour(cr); cg = gsr_finite_colour(cg); cbl = gsr_finite_colour(cbl);
}

// the splat's 128-byte row (position + 48 colour halves, two sectors) -> colour
__device__ __forceinline__ void gsr_splat_colour_from_row(const GsrFrame& f, const uint4* __restrict__ colrow, uint32_t idx,
                                                          float& cr, float& cg, float& cbl)
{
    const uint4* row = colrow + (size_t)idx * 8;
    const uint4 g = row[0];
    uint4 cw[6];
    const int nchunk = f.sh_order == 0 ? 1 : (f.sh_order == 1 ? 2 : (f.sh_order == 2 ? 4 : 6));
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        cw[c] = make_uint4(0, 0, 0, 0);
        if (c < nchunk) cw[c] = row[1 + c];
    }
    const float px = __uint_as_float(g.x), py = __uint_as_float(g.y), pz = __uint_as_float(g.z);
    const float x = (px - f.origin[0]) + f.origin[0];
    const float y = (py - f.origin[1]) + f.origin[1];
    const float z = (pz - f.origin[2]) + f.origin[2];
    gsr_splat_colour(f, cw, x, y, z, cr, cg, cbl);
}

// K1: the per-splat vertex stage.
//   in : geoA, geoB, col (SoA, coalesced 16 B/lane)
//   out: rec[i] (48 B), key[i] (f32 distance^2 bits), val[i] = (i, rect)
// Two halves.  gsr_k1_front: key, clip test, centre, and -- band layout, from a cheap bound of the quad's extent -- whether the
// splat lies far from this rank's band of tile rows.  gsr_k1_back: covariance chain, tile rect, occlusion cull, colour, record.
// (Measured and dropped: culling BEFORE the covariance chain from the cheap bound, with the survivors compacted -- per
// workgroup, per wave, or into dense arrays by a second kernel.  The chain is not what this kernel waits for; every variant
// replaced one streaming pass by scattered 16-128-byte accesses, which run at a third of the streaming rate here, and lost.)
struct GsrK1Front {
    uint32_t kb;        // sort key (distance^2 bits, range-reduced)
    float x, y, z;      // fl32(P - origin) + origin
    float cx, cy, opacity;
    float zw;           // window depth of the quad (depth-tested frames only)
    bool keep;          // passes the clip tests
    bool far;           // ... but cannot touch this rank's rows / lies behind every horizon it can reach
};

__device__ __forceinline__ GsrK1Front
gsr_k1_front(const GsrFrame& f, const float4 a, const uint4 b, float* __restrict__ zwin_i, uint32_t slab_key = 0xffffffffu)
{
    GsrK1Front o;
    const float px = a.x, py = a.y, pz = a.z;
    o.opacity = a.w;
    // sort key: un-offset P vs camera (src/GSplatRenderer.C:197-201).  distance^2 >= 0: its IEEE bits
    // are monotone.  The host bounds them for this frame from the cloud's bounding box
    // (key_min/key_max), so the sort only has to cover key_max - key_min.
    {
        float dx = px - f.cam[0], dy = py - f.cam[1], dz = pz - f.cam[2];
        float k = gsr_fma(dz, dz, gsr_fma(dy, dy, dx * dx));
        uint32_t kb = __builtin_bit_cast(uint32_t, k);
        kb = kb < f.key_min ? f.key_min : (kb > f.key_max ? f.key_max : kb);
        o.kb = kb - f.key_min;
    }
    // fl32(P - origin) + origin  (src/GSplatRenderer.C:459-461, shader :201-202)
    const float x = (px - f.origin[0]) + f.origin[0];
    const float y = (py - f.origin[1]) + f.origin[1];
    const float z = (pz - f.origin[2]) + f.origin[2];
    o.x = x; o.y = y; o.z = z;

    const float tvx = aff4(&f.ov[0], x, y, z);
    const float tvy = aff4(&f.ov[4], x, y, z);
    const float tvz = aff4(&f.ov[8], x, y, z);
    const float ftvy = -tvy;  // flipYMatrix (:204-207)
    const float clx = aff4(&f.pr[0], tvx, ftvy, tvz);
    const float cly = aff4(&f.pr[4], tvx, ftvy, tvz);
    const float clz = aff4(&f.pr[8], tvx, ftvy, tvz);
    const float clw = aff4(&f.pr[12], tvx, ftvy, tvz);

    // w<=0 (:209-214); near/far clip of a constant-z quad; alpha = e*opacity <= opacity
    // can never reach 1/255 when opacity < 1/255 (e <= 1), so those splats draw nothing.
    o.keep = (clw > 0.0f) && !(clz < -clw || clz > clw) && (o.opacity >= (1.0f / 255.0f));
    // front-slab frames: phase 1 draws the splats up to the slab key, phase 2 the ones beyond it (ties at the key: phase 1)
    if (f.phase == 1) o.keep = o.keep && o.kb <= slab_key;
    if (f.phase == 2) o.keep = o.keep && o.kb > slab_key;
    o.far = false;
    o.cx = 0.0f; o.cy = 0.0f; o.zw = 0.0f;
    if (o.keep) {
        const float ndcx = clx / clw;
        const float ndcy = (-cly) / clw;
        const float cx = gsr_fma(ndcx, 0.5f, 0.5f) * f.W;
        const float cy = gsr_fma(ndcy, 0.5f, 0.5f) * f.H;
        o.cx = cx; o.cy = cy;
        // every corner carries the centre's z and w: one window depth per quad (depth range 0..1)
        if (zwin_i) { o.zw = gsr_fma(clz / clw, 0.5f, 0.5f); *zwin_i = o.zw; }

        // A cheap UPPER BOUND of the quad's half extent in pixels, without the covariance chain (lambda1 <= trace(cov2d) <=
        // |J|_F^2 |V|_2^2 |O|_2^2 |diag(s) R^T|_F^2 + 0.6; h <= 2 sqrt2 s1 1.0001 + 0.01).  Conservative, so what it
        // decides never changes what is drawn.  Band layout: most splats lie far from this rank's band of tile rows.
        if (f.shard_rpb > 0) {
            const float mf = gsr_h2f(b.w >> 16);   // >= |diag(scale) R^T|_F, from k_pack
            const float mf2 = mf * mf;
            const float tzb = aff4(&f.vw[8], x, y, z);
            const float jz = f.focal / tzb;
            const float trb = jz * jz * (2.0f + f.limx * f.limx + f.limy * f.limy) * f.sigma_vo2 * mf2 * 1.001f + 0.6f;
            const float hb = 2.8313f * __builtin_fminf(__builtin_sqrtf(2.0f * trb), 4096.0f) + 0.02f;
            if (hb < 1.0e9f) {   // (false for NaN / inf: those take the full path and its finite-covariance rule)
                const float lo_px = cy - hb - 0.5f, hi_px = cy + hb - 0.5f;
                if (f.shard_rpb > 0) {
                    const int band_lo = f.shard_index * f.shard_rpb * GSR_TILE_PX;
                    const int band_hi = band_lo + f.shard_rpb * GSR_TILE_PX - 1;
                    if (hi_px < (float)band_lo || lo_px > (float)band_hi) o.far = true;
                }
            }
        }
    }
    return o;
}

// returns the packed tile rect (GSR_RECT_EMPTY: the splat draws nothing here); writes the record of a splat that draws
__device__ __forceinline__ uint32_t
gsr_k1_back(const GsrFrame& f, uint32_t i, uint32_t cap, const GsrK1Front& o, const uint4 b, const uint4* __restrict__ col,
            GsrRecord* __restrict__ rec, int lazy, const float* __restrict__ hpyr, const float* __restrict__ dpyr /* NULL: no depth culling */,
            const float* __restrict__ dpyrc)
{
    uint32_t out_rect = GSR_RECT_EMPTY;
    const float x = o.x, y = o.y, z = o.z, cx = o.cx, cy = o.cy, opacity = o.opacity;
    const float sx = gsr_h2f(b.x & 0xffffu), sy = gsr_h2f(b.x >> 16), sz = gsr_h2f(b.y & 0xffffu);
    const float qi = gsr_h2f(b.y >> 16), qj = gsr_h2f(b.z & 0xffffu), qk = gsr_h2f(b.z >> 16);
    const float qr = gsr_h2f(b.w & 0xffffu);
    float ex = 1.0f, ey = 0.0f, s1 = 0.0f, s2 = 0.0f;
    const bool finite = gsr_covariance_axes(f, f.ob, x, y, z, sx, sy, sz, qi, qj, qk, qr, ex, ey, s1, s2);
    // conservative bbox of the part of the quad where alpha can reach 1/255 (|q| <= rq <= 2)
    // (native log / sqrt: the bbox is not parity-relevant, only conservative -- the margins dwarf their 1-ulp error)
    const float rq = (f.flags & GSR_FLAG_NO_ALPHA_RADIUS) ? 2.0f : gsr_support_radius_fast(opacity);
    const float hx = gsr_fma(rq * gsr_fma(s1, __builtin_fabsf(ex), s2 * __builtin_fabsf(ey)), 1.0001f, 0.01f);
    const float hy = gsr_fma(rq * gsr_fma(s1, __builtin_fabsf(ey), s2 * __builtin_fabsf(ex)), 1.0001f, 0.01f);

    // pixel range of the conservative bbox -> tile rect
    const float xlo = cx - hx - 0.5f, xhi = cx + hx - 0.5f;
    const float ylo = cy - hy - 0.5f, yhi = cy + hy - 0.5f;
    const float wm1 = (float)(f.width - 1), hm1 = (float)(f.height - 1);
    if (finite && xhi >= 0.0f && xlo <= wm1 && yhi >= 0.0f && ylo <= hm1) {
        const int i0 = (int)__builtin_ceilf(__builtin_fmaxf(xlo, 0.0f));
        const int i1 = (int)__builtin_floorf(__builtin_fminf(xhi, wm1));
        const int j0 = (int)__builtin_ceilf(__builtin_fmaxf(ylo, 0.0f));
        const int j1 = (int)__builtin_floorf(__builtin_fminf(yhi, hm1));
        if (i1 >= i0 && j1 >= j0) { const int g4 = 4 + f.rect_shift; out_rect = gsr_pack_rect(i0 >> g4, j0 >> g4, i1 >> g4, j1 >> g4); }
    }
    // a splat none of whose tiles belong to this context's row shard is dropped here: it costs no
    // colour fetch, no record and (sentinel key) no sorting
    if (out_rect != GSR_RECT_EMPTY && gsr_rect_tiles(out_rect, GsrShard{f.shard_index, f.shard_count, f.shard_rpb, f.rect_shift}) == 0)
        out_rect = GSR_RECT_EMPTY;
    // Occlusion culling against the previous frame's depth horizons (k_blend.h, k_sum_work): a tile that went opaque at some
    // depth needs nothing behind it.  A splat whose key lies beyond the horizon of EVERY tile its rect reaches (widened by the
    // frame's dilation radius: the view moves between frames) is dropped here -- no colour, no record, no sorting, no binning.
    // Every list therefore holds, for each of its tiles, all the splats in front of that tile's horizon; a tile that had to
    // look further than its horizon reports the frame, which is then rendered again without culling (k_sum_work).
    // (The pyramid is read through the cache: at most four gathers, usually of one or two lines.)
    // (depth-tested frames: "beyond every horizon" is half of the verdict -- see the depth clause below)
    bool beyond = false, no_horizon = true;     // (no finite horizon over the tiles the rect reaches, or no horizons in this frame at all)
    if (hpyr && out_rect != GSR_RECT_EMPTY) {
        const int g = f.rect_shift;
        const int x0 = (int)(out_rect & 255u) << g, y0 = (int)((out_rect >> 8) & 255u) << g;
        const int x1 = (((int)((out_rect >> 16) & 255u) + 1) << g) - 1, y1 = (((int)(out_rect >> 24) + 1) << g) - 1;
        const int r = f.cull_dilate;
        const float h = gsr_pyr_max(hpyr, f.pyr_off, f.tiles_x, max(x0 - r, 0), max(y0 - r, 0), min(x1 + r, f.tiles_x - 1), min(y1 + r, f.tiles_y - 1));
        beyond = o.kb > gsr_horizon_key(h, f.key_min, f.key_max);
        no_horizon = !(h < 3.0e38f);
        if (beyond && !dpyr) out_rect = GSR_RECT_EMPTY;
    }
    // Depth-tested frames: the quad carries ONE window depth (o.zw) and a fragment survives iff zw <= depth[pixel]
    // (src/GSplatRenderer.C:595-610).  Two rules (k_cluster.h: the two tile-max pyramids of the depth buffer):
    //  * beyond the largest depth the opaque pass left under every tile the rect reaches, no fragment survives: dropped, whatever the
    //    horizons say;
    //  * the horizons speak for the tiles' UNCOVERED pixels only (a pixel under opaque geometry may never saturate): a splat beyond
    //    them is dropped only if it is also behind everything under the COVERED pixels of its tiles -- those get what lies in front of
    //    the geometry exactly, with no prediction to verify.  (Culling by depth INSIDE the horizons was tried first and spoilt them: k_tile_pass
    //    places a horizon a quarter + 1024 entries down the list, a list cut at the geometry is "too short" for that, and the fallback
    //    pushes the horizon out by 5 % per frame for good -- 13 k surviving clusters became 45 k on C4 with a sphere under 30 % of the frame.)
    if (dpyr && out_rect != GSR_RECT_EMPTY) {
        const int g = f.rect_shift;
        const int x0 = (int)(out_rect & 255u) << g, y0 = (int)((out_rect >> 8) & 255u) << g;
        const int x1 = min((((int)((out_rect >> 16) & 255u) + 1) << g) - 1, f.tiles_x - 1), y1 = min((((int)(out_rect >> 24) + 1) << g) - 1, f.tiles_y - 1);
        // (the first rule only where no finite horizon applies: INSIDE a horizon nothing is dropped by depth -- k_tile_pass places the next
        //  horizon a quarter + 1024 entries down the list, a list cut at the geometry is "too short" for that, and its fallback pushes the
        //  horizon out by 5 % per frame for good: measured, 13 k surviving clusters became 45 k)
        if (no_horizon) { if (o.zw > gsr_dpyr_max(dpyr, f.pyr_off, f.tiles_x, x0, y0, x1, y1)) out_rect = GSR_RECT_EMPTY; }
        // (the covered depths of the tiles that were NOT classic in the frame that left the horizons, looked up over the rect widened like
        //  the horizon look-up: a tile's status is as old as its horizon.  No pyrc: no depth clause at all -- phase 2 of a front-slab
        //  frame, whose "horizons" are 0 for the tiles phase 1 FINISHED and +inf for the others)
        else if (beyond) {
            const int r = f.cull_dilate;
            if (!dpyrc || o.zw > gsr_dpyr_max(dpyrc, f.pyr_off, f.tiles_x, max(x0 - r, 0), max(y0 - r, 0), min(x1 + r, f.tiles_x - 1), min(y1 + r, f.tiles_y - 1)))
                out_rect = GSR_RECT_EMPTY;
        }
    }
    if (out_rect != GSR_RECT_EMPTY) {
        // colour: Cd, optionally + SH (:224, :244-274) -- or left PENDING for the lazy colour pass (k_colour.h)
        float cr, cg, cbl;
        if (lazy) {
            cr = __builtin_bit_cast(float, GSR_COLOUR_PENDING); cg = 0.0f; cbl = 0.0f;
        } else {
            uint4 cw[6];
            cw[0] = col[i];
            const int nchunk = f.sh_order == 0 ? 1 : (f.sh_order == 1 ? 2 : (f.sh_order == 2 ? 4 : 6));
#pragma unroll
            for (int c = 1; c < 6; ++c) {
                cw[c] = make_uint4(0, 0, 0, 0);
                if (c < nchunk) cw[c] = col[(size_t)c * cap + i];
            }
            gsr_splat_colour(f, cw, x, y, z, cr, cg, cbl);
        }
        // contract v2: the quad-local coordinate as two affine forms scaled by kappa = sqrt(log2 e)
        const float k1 = (1.0f / s1) * GSR_KAPPA, k2 = (1.0f / s2) * GSR_KAPPA;
        float4* dst = reinterpret_cast<float4*>(rec + i);
        dst[0] = make_float4(cx, cy, hx, hy);
        dst[1] = make_float4(ex * k1, ey * k1, -(ey * k2), ex * k2);
        dst[2] = make_float4(cr, cg, cbl, gsr_log2_opacity(opacity));
    }
    return out_rect;
}

struct GsrK1Scatter {
    uint32_t* key;                 // [BK_BUCKETS * BK_CAP] bucket regions, or NULL: K1 leaves the bucket pass to k_bucket_scatter*
    uint2* val;
    uint32_t* cnt;                 // [BK_BUCKETS * BK_STRIDE] keys per bucket (cleared by k_cluster_cull)
    uint32_t* failed;              // set when a bucket's region is full
    const uint32_t* range_dev;     // { lo, shift } on the device (front-slab phases), or NULL: the two below
    uint32_t lo;
    int32_t shift;
};

// One wavefront per surviving cluster (k_cluster.h): workgroup-iteration k takes the clusters of rank 4k .. 4k+3 of
// k_cluster_cull's ordered list, so its 256 slots are in storage order like the list itself.  The grid is sized from the
// previous frame's survivor count (gsr_api.hip); a frame that keeps more simply loops.
// (6 waves per SIMD = 80 VGPRs: with the colours evaluated here, 7 waves (72 VGPRs) spill 12 dwords: 38.2 -> 34.5 us on a culled C4 frame)
#ifndef GSR_K1_WAVES_PER_EU
#define GSR_K1_WAVES_PER_EU 6
#endif
// The LAZY instantiation (colours left pending: unculled frames of scenes with depth complexity) has no SH evaluation in it and needs
// 52 VGPRs instead of 80: it runs at 8 waves per SIMD.  K1 is bound by resident workgroups x the ~10 us a workgroup lives (12.7
// generations of 1536 workgroups x 10.4 us = the 133 us of an unculled C4 frame, tools/kprof.py), so occupancy is its lever.
#ifndef GSR_K1_WAVES_PER_EU_LAZY
#define GSR_K1_WAVES_PER_EU_LAZY 8
#endif
// (DEPTH: the frame is depth-tested -- window depths are written and the depth rules of gsr_k1_back apply.  A template parameter, not a
//  run-time pointer test: the plain shading instantiation is the headline frame's K1, and its 80 registers have no room for the extra state)
template <bool LAZY, bool DEPTH>
__device__ __forceinline__ void
gsr_k1_body(uint32_t n, uint32_t cap, const GsrFrame& f,
             const float4* __restrict__ geoA, const uint4* __restrict__ geoB, const uint4* __restrict__ col,
             GsrRecord* __restrict__ rec, uint32_t* __restrict__ key, uint2* __restrict__ val,
             float* __restrict__ zwin /* NULL unless the frame is depth-tested */,
             const float* __restrict__ hpyr /* depth-horizon pyramid, or NULL: no occlusion culling */,
             uint32_t* __restrict__ blk_cnt /* [workgroup-iterations] splats of each that stay */,
             const uint32_t* __restrict__ cseg, const uint32_t* __restrict__ ccnt, uint32_t ngroups, uint32_t cper /* k_cluster_cull's output */,
             uint32_t* __restrict__ d_counts /* [0] = slots K1 filled (256 per workgroup-iteration), [1] = surviving clusters, [2] = "the small-frame sort
                                                gave a bucket up" (cleared here, set by k_bucket_scatter / k_radix_local) */,
             GsrK1Scatter sc /* small-frame sort: the kept keys go straight into its buckets (sc.key != NULL) */,
             uint32_t* __restrict__ zero_n /* the count of sorted splats, cleared here */,
             const uint32_t* __restrict__ order /* position-keyed order (GSR_OPT_SORT_CACHE = 2): slot j holds splat order[j], the splats
                                                   are walked nearest first and leave already sorted; NULL = storage order */,
             const uint32_t* __restrict__ slab /* front-slab frames (f.phase != 0): [0] = the slab key (k_slab_pick) */,
             GsrDepthCull dc /* depth-tested frames: the opaque pass's tile-max depth pyramid (k_cluster.h), pyr = NULL otherwise */)
{
    static_assert(GSR_K1_THREADS == 4 * GSR_CLUSTER, "a K1 workgroup is four clusters");
    __shared__ uint32_t s_inc[CC_MAX_GROUPS];
    __shared__ uint32_t s_wcnt[2][GSR_K1_THREADS / 64];
    __shared__ uint32_t s_scan[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    KPROFB(4, 0, gridDim.x / 2)
    KPROF_BLK_BEGIN
    const uint32_t nsurv = cc_prefix_to_lds(ccnt, ngroups, s_inc, s_scan);
    KPROFB(4, 1, gridDim.x / 2)
    const uint32_t niter = (nsurv + 3u) / 4u;
    if (blockIdx.x == 0) {
        // (d_counts[2], "the small-frame sort gave a bucket up", and the bucket counters were cleared by k_cluster_cull)
        if (threadIdx.x == 0) { d_counts[0] = niter * (uint32_t)GSR_K1_THREADS; d_counts[1] = nsurv; if (zero_n) *zero_n = 0u; }
    }
    uint32_t sc_lo = sc.lo;
    int sc_shift = sc.shift;
    if (sc.key && sc.range_dev) { sc_lo = sc.range_dev[0]; sc_shift = (int)sc.range_dev[1]; }
    const uint32_t slab_key = (f.phase != 0 && slab) ? slab[0] : 0xffffffffu;
    const float* const dpyr = (DEPTH && dc.pyr && *dc.active != 0u) ? dc.pyr : (const float*)nullptr;   // (uniform: nothing changes under a cleared depth buffer)
    int par = 0;
    for (uint32_t k = blockIdx.x; k < niter; k += gridDim.x, par ^= 1) {
        const uint32_t rank = 4u * k + (uint32_t)wave;
        uint32_t out_rect = GSR_RECT_EMPTY, kb = 0, i = 0;
        if (rank < nsurv) {                                   // (wave-uniform)
            const uint32_t cl = cc_find_cluster(s_inc, ngroups, rank, cseg, cper);
            i = cl * (uint32_t)GSR_CLUSTER + (uint32_t)lane;
            const bool exists = i < n;
            if (order && exists) i = order[i];
            if (exists) {
                // geoA and geoB are fetched together; colour only once the splat is known to be needed (fetching it up front
                // was measured slower: the bytes wasted on culled splats cost more than the second round trip)
                const float4 a = geoA[i];
                const uint4 b = geoB[i];
#ifdef GSR_KPROF
                if (k == blockIdx.x && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); g_kprof[4][2] = wall_clock64(); }
#endif
                const GsrK1Front o = gsr_k1_front(f, a, b, (DEPTH && zwin) ? zwin + i : nullptr, slab_key);
                if (o.keep && !o.far) out_rect = gsr_k1_back(f, i, cap, o, b, col, rec, LAZY ? 1 : 0, hpyr, DEPTH ? dpyr : (const float*)nullptr, DEPTH ? dc.pyrc : (const float*)nullptr);
                kb = o.kb;
            }
        }
        // The keys and payloads of the splats that stay leave compacted PER WORKGROUP-ITERATION, in thread (= storage) order, at
        // the head of its 256 slots, with their number in blk_cnt: the depth sort's first pass gathers those prefixes (k_sort.h,
        // GATHER) instead of reading one key per splat, and, the slots being in storage order, equal keys leave the stable
        // sort in storage order.
        if (k == blockIdx.x) { KPROFB(4, 3, gridDim.x / 2) }
        const bool stays = out_rect != GSR_RECT_EMPTY;
        if (sc.key) {                                  // (uniform)
            // The small-frame sort's bucket pass, here: a key's place in its bucket is what one atomic on the bucket's counter returns
            // (k_sort.h; order inside a bucket is k_radix_local's business: it puts ties back into storage order).
            // Bucket = min((key - lo) >> shift, BK_BUCKETS - 1).  Nothing else reads K1's slots in such a frame: no compaction.
            if (stays) {
                const uint32_t t = kb > sc_lo ? (kb - sc_lo) >> sc_shift : 0u;
                const uint32_t d = t < (uint32_t)(BK_BUCKETS - 1) ? t : (uint32_t)(BK_BUCKETS - 1);
                const uint32_t p = atomicAdd(&sc.cnt[(size_t)d * BK_STRIDE], 1u);
                if (p < (uint32_t)BK_CAP) { sc.key[(size_t)d * BK_CAP + p] = kb; sc.val[(size_t)d * BK_CAP + p] = make_uint2(i, out_rect); }
                else *sc.failed = 1u;             // the bucket's region is full: the prediction missed badly
            }
            continue;
        }
        const unsigned long long bal = __ballot(stays);
        if (lane == 0) s_wcnt[par][wave] = (uint32_t)__builtin_popcountll(bal);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < GSR_K1_THREADS / 64; ++w) { const uint32_t c = s_wcnt[par][w]; before += w < wave ? c : 0u; total += c; }
        if (stays) {
            const uint32_t pos = k * (uint32_t)GSR_K1_THREADS + before + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
            key[pos] = kb;
            val[pos] = make_uint2(i, out_rect);   // sort payload: storage index + tile rect
        }
        if (threadIdx.x == 0) blk_cnt[k] = total;
        if (k == blockIdx.x) { KPROFB(4, 4, gridDim.x / 2) }
    }
    KPROF_BLK_END(4, (blockIdx.x < niter ? 1u : 0u) + (niter > blockIdx.x ? (niter - 1u - blockIdx.x) / gridDim.x : 0u))
}

#define GSR_K1_PARAMS uint32_t n, uint32_t cap, GsrFrame f, const float4* __restrict__ geoA, const uint4* __restrict__ geoB, const uint4* __restrict__ col, \
                      GsrRecord* __restrict__ rec, uint32_t* __restrict__ key, uint2* __restrict__ val, float* __restrict__ zwin, const float* __restrict__ hpyr, \
                      uint32_t* __restrict__ blk_cnt, const uint32_t* __restrict__ cseg, const uint32_t* __restrict__ ccnt, uint32_t ngroups, uint32_t cper, \
                      uint32_t* __restrict__ d_counts, GsrK1Scatter sc, uint32_t* __restrict__ zero_n, const uint32_t* __restrict__ order, const uint32_t* __restrict__ slab, \
                      GsrDepthCull dc
#define GSR_K1_ARGS n, cap, f, geoA, geoB, col, rec, key, val, zwin, hpyr, blk_cnt, cseg, ccnt, ngroups, cper, d_counts, sc, zero_n, order, slab, dc
// the two entry points: colours evaluated here (eager: 80 VGPRs, 6 waves per SIMD) or left pending (52 VGPRs, 8 waves per SIMD)
__global__ void __launch_bounds__(GSR_K1_THREADS) __attribute__((amdgpu_waves_per_eu(GSR_K1_WAVES_PER_EU)))
k_preprocess(GSR_K1_PARAMS) { gsr_k1_body<false, false>(GSR_K1_ARGS); }
__global__ void __launch_bounds__(GSR_K1_THREADS) __attribute__((amdgpu_waves_per_eu(GSR_K1_WAVES_PER_EU_LAZY)))
k_preprocess_lazy(GSR_K1_PARAMS) { gsr_k1_body<true, false>(GSR_K1_ARGS); }
// ... and their depth-tested twins (the shading one is left to the register allocator: it carries the depth rules on top of the SH evaluation)
__global__ void __launch_bounds__(GSR_K1_THREADS)
k_preprocess_depth(GSR_K1_PARAMS) { gsr_k1_body<false, true>(GSR_K1_ARGS); }
__global__ void __launch_bounds__(GSR_K1_THREADS) __attribute__((amdgpu_waves_per_eu(GSR_K1_WAVES_PER_EU_LAZY)))
k_preprocess_lazy_depth(GSR_K1_PARAMS) { gsr_k1_body<true, true>(GSR_K1_ARGS); }

// upload time: per-workgroup partial bounding boxes of the positions (finished on the host), from the raw float[3] array in the arena
__global__ void __launch_bounds__(256)
k_bbox_partials(const float* __restrict__ P, uint32_t n, float* __restrict__ partial /*[grid][6]*/)
{
    __shared__ float s[4][6];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    bool finite = true;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float p[3] = {P[3 * (size_t)i], P[3 * (size_t)i + 1], P[3 * (size_t)i + 2]};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            finite = finite && (__builtin_fabsf(p[k]) < 3.0e38f);   // false for inf and NaN
            lo[k] = __builtin_fminf(lo[k], p[k]);
            hi[k] = __builtin_fmaxf(hi[k], p[k]);
        }
    }
    if (!finite) { lo[0] = -__builtin_inff(); hi[0] = __builtin_inff(); }   // poison: host falls back to full keys
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            lo[k] = __builtin_fminf(lo[k], __shfl_down(lo[k], d, 64));
            hi[k] = __builtin_fmaxf(hi[k], __shfl_down(hi[k], d, 64));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) { s[wave][k] = lo[k]; s[wave][3 + k] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? __builtin_fminf(v, s[w][threadIdx.x]) : __builtin_fmaxf(v, s[w][threadIdx.x]);
        partial[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

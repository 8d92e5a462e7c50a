// k_wire.h -- wireframe overlay: the outline of every splat's +-2 quad, colour Cd, opaque,
// nearest line wins.  Replaces the reference's wire program
// (/root/reference/gsplat_plugin/shaders/GSplatShaderSource.h:22-110, 8-vertex line list per splat built
// in src/GR_GSplat.C:374-421 and drawn in :477-483).  Note what that program does NOT do, and neither
// does this: no GSplatOrigin round trip, no object matrix in the covariance, no SH.
// A debug overlay, not on the beauty path: one thread per splat walks its four edges.
#pragma once
#include "gsr_device.h"
#include "k_preprocess.h"

#define GSR_WIRE_EMPTY 0xffffffffffffffffull

// line rule (the contract's stand-in for GL's diamond-exit rule): along the major axis every pixel
// centre in [min, max) of the segment gets one fragment; the minor coordinate is floor() of the line
// evaluated at that centre
__device__ __forceinline__ void gsr_wire_edge(float x0, float y0, float x1, float y1, int width, int height,
                                              unsigned long long frag, unsigned long long* __restrict__ zbuf)
{
    const float dx = x1 - x0, dy = y1 - y0;
    if (!(__builtin_fabsf(dx) < 3.0e38f) || !(__builtin_fabsf(dy) < 3.0e38f)) return;   // inf / NaN corner
    const bool xmajor = __builtin_fabsf(dx) >= __builtin_fabsf(dy);
    const float m0 = xmajor ? x0 : y0, m1 = xmajor ? x1 : y1;     // major-axis endpoints
    const float n0 = xmajor ? y0 : x0;                            // minor-axis start
    const float dm = xmajor ? dx : dy, dn = xmajor ? dy : dx;
    if (dm == 0.0f) return;                                        // degenerate (zero-length) edge
    const float lo = __builtin_fminf(m0, m1), hi = __builtin_fmaxf(m0, m1);
    const int mmax = (xmajor ? width : height) - 1, nmax = (xmajor ? height : width) - 1;
    const float flo = __builtin_ceilf(lo - 0.5f), fhi = __builtin_ceilf(hi - 0.5f) - 1.0f;
    if (!(fhi >= 0.0f && flo <= (float)mmax)) return;
    const int i0 = (int)__builtin_fmaxf(flo, 0.0f), i1 = (int)__builtin_fminf(fhi, (float)mmax);
    for (int i = i0; i <= i1; ++i) {
        const float t = (((float)i + 0.5f) - m0) / dm;
        const float nv = __builtin_floorf(gsr_fma(t, dn, n0));
        if (!(nv >= 0.0f && nv <= (float)nmax)) continue;
        const int j = (int)nv;
        const size_t pix = xmajor ? ((size_t)j * width + i) : ((size_t)i * width + j);
        atomicMin(&zbuf[pix], frag);
    }
}

__global__ void __launch_bounds__(256)
k_wire_splats(uint32_t n, GsrFrame f, const float4* __restrict__ geoA, const uint4* __restrict__ geoB,
              unsigned long long* __restrict__ zbuf,
              const uint32_t* __restrict__ perm /* storage slot -> index in the upload (spatially ordered storage), or NULL */)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = geoA[i];
    const uint4 b = geoB[i];
    const float x = a.x, y = a.y, z = a.z;                        // no origin offset in the wire program
    const float tvx = aff4(&f.ov[0], x, y, z), tvy = aff4(&f.ov[4], x, y, z), tvz = aff4(&f.ov[8], x, y, z);
    const float ftvy = -tvy;
    const float clx = aff4(&f.pr[0], tvx, ftvy, tvz), cly = aff4(&f.pr[4], tvx, ftvy, tvz);
    const float clz = aff4(&f.pr[8], tvx, ftvy, tvz), clw = aff4(&f.pr[12], tvx, ftvy, tvz);
    // all eight vertices share the centre's z and w: GL clips the whole outline or none of it
    if (!(clw > 0.0f) || clz < -clw || clz > clw) return;
    const float cx = gsr_fma(clx / clw, 0.5f, 0.5f) * f.W;
    const float cy = gsr_fma((-cly) / clw, 0.5f, 0.5f) * f.H;
    const float zw = gsr_fma(clz / clw, 0.5f, 0.5f);
    const float sx = gsr_h2f(b.x & 0xffffu), sy = gsr_h2f(b.x >> 16), sz = gsr_h2f(b.y & 0xffffu);
    const float qi = gsr_h2f(b.y >> 16), qj = gsr_h2f(b.z & 0xffffu), qk = gsr_h2f(b.z >> 16), qr = gsr_h2f(b.w & 0xffffu);
    const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float ex, ey, s1, s2;
    if (!gsr_covariance_axes(f, ident, x, y, z, sx, sy, sz, qi, qj, qk, qr, ex, ey, s1, s2)) return;   // non-finite covariance
    // corner(sx, sy) = c + sx*(2 s1 e) + sy*(2 s2 e_perp), e_perp = (-ey, ex)
    const float ax = (2.0f * s1) * ex, ay = (2.0f * s1) * ey;
    const float bx = (2.0f * s2) * (-ey), by = (2.0f * s2) * ex;
    const float c0x = (cx - ax) - bx, c0y = (cy - ay) - by;       // (-2,-2)
    const float c1x = (cx + ax) - bx, c1y = (cy + ay) - by;       // (+2,-2)
    const float c2x = (cx + ax) + bx, c2y = (cy + ay) + by;       // (+2,+2)
    const float c3x = (cx - ax) + bx, c3y = (cy - ay) + by;       // (-2,+2)
    // nearest fragment wins, earlier splat on equal depth (GL_LESS in draw order): min of (depth bits, index) -- the index in the
    // UPLOAD, which is the order the reference's line list is built and drawn in (src/GR_GSplat.C:374-421), not the storage slot:
    // under an orthographic camera with a distant far plane whole groups of outlines share their depth bits
    const unsigned long long frag = ((unsigned long long)__builtin_bit_cast(uint32_t, zw) << 32) | (unsigned long long)(perm ? perm[i] : i);
    gsr_wire_edge(c0x, c0y, c1x, c1y, f.width, f.height, frag, zbuf);   // vertices 0-1
    gsr_wire_edge(c1x, c1y, c2x, c2y, f.width, f.height, frag, zbuf);   // 2-3
    gsr_wire_edge(c2x, c2y, c3x, c3y, f.width, f.height, frag, zbuf);   // 4-5
    gsr_wire_edge(c3x, c3y, c0x, c0y, f.width, f.height, frag, zbuf);   // 6-7
}

__global__ void __launch_bounds__(256)
k_wire_resolve(const unsigned long long* __restrict__ zbuf, size_t npix, const uint4* __restrict__ col0,
               float4* __restrict__ out, const uint32_t* __restrict__ inv /* index in the upload -> storage slot, or NULL */,
               int over /* 1 = wire-over display: only the pixels an outline covers are written, the frame underneath stays */)
{
    const size_t p = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (p >= npix) return;
    const unsigned long long v = zbuf[p];
    if (over && v == GSR_WIRE_EMPTY) return;
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (v != GSR_WIRE_EMPTY) {
        const uint32_t idx = (uint32_t)(v & 0xffffffffull);
        const uint4 c = col0[inv ? inv[idx] : idx];               // chunk 0 starts with Cd.rgb (f16)
        o = make_float4(gsr_h2f(c.x & 0xffffu), gsr_h2f(c.x >> 16), gsr_h2f(c.y & 0xffffu), 1.0f);
    }
    out[p] = o;
}

// inv[perm[j]] = j
__global__ void __launch_bounds__(256)
k_invert_perm(const uint32_t* __restrict__ perm, uint32_t n, uint32_t* __restrict__ inv)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j < n) inv[perm[j]] = j;
}

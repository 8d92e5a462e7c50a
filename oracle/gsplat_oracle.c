/*
 * gsplat_oracle.c -- CPU ORACLE (test infrastructure; see gsplat_oracle.h).
 *
 * Plain C restatement of the reference's per-frame render path.  Every block
 * cites the reference lines it follows (paths relative to
 * /root/reference/gsplat_plugin).  Build: see oracle/Makefile
 * (-O2 -ffp-contract=off -mfma: only explicit fmaf() fuses).
 */
#include "gsplat_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define M4(m, r, c) ((m)[(c) * 4 + (r)])

/* ------------------------------------------------------------------------- */
/* binary16 <-> binary32.  The reference quantises Cd/scale/orient/SH through
 * HDK fpreal16 constructors (src/GR_GSplat.C:315-318,345-367), i.e. IEEE
 * round-to-nearest-even with overflow to infinity.                          */

float gso_half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal half -> normal float */
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            man &= 0x3ffu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

uint16_t gso_float_to_half(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
    if (ax >= 0x477ff000u) /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    if (ax < 0x33000001u) /* <= 2^-25 rounds to zero (tie at 2^-25 -> even = 0) */
        return (uint16_t)sign;
    int e = (int)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7fffffu) | 0x800000u; /* 24-bit significand */
    int shift;                                   /* bits to drop */
    uint32_t hexp;
    if (e < -14) { /* subnormal half */
        shift = 13 + (-14 - e);
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t kept = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t half_ulp = 1u << (shift - 1);
    if (rem > half_ulp || (rem == half_ulp && (kept & 1u)))
        ++kept;
    uint32_t out;
    if (hexp == 0)
        out = kept; /* may carry into exponent 1: correct */
    else
        out = ((hexp - 1) << 10) + kept; /* kept has the implicit bit at 0x400 */
    return (uint16_t)(sign | out);
}

/* ------------------------------------------------------------------------- */
/* The contract's exp().  GLSL exp() (shaders/GSplatShaderSource.h:307) has
 * no bit-level definition; the contract fixes one so that every threshold
 * decision (alpha < 1/255) is reproducible.  Classic range reduction
 * x = k*ln2 + r, |r| <= ln2/2, degree-5 polynomial for (exp(r)-1-r)/r^2
 * (coefficients: the widely published single-precision minimax set),
 * ~1 ulp.  Valid for -80 <= x <= 0 (the path only needs [-8, 0]).          */
float gso_expf(float x)
{
    float kf = rintf(x * 1.44269504088896341f);
    float r = fmaf(kf, -0.693359375f, x);
    r = fmaf(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    /* y * 2^k by exponent-field add; y in [0.70, 1.42], k in [-116, 0] */
    int32_t k = (int32_t)kf;
    uint32_t bits;
    memcpy(&bits, &y, 4);
    bits += (uint32_t)k << 23;
    memcpy(&y, &bits, 4);
    return y;
}

/* Contract v2: the fragment stage evaluates the gaussian as 2^(-|kappa q|^2) with kappa^2 = log2(e) folded into the
 * quad axes (vertex stage), so exp() becomes a base-2 exponential whose range reduction is EXACT
 * (r = x - rint(x)) -- 8 instead of 13 operations per fragment on the GPU.  x <= 0, |x| < 2^22:
 * s = x + 1.5*2^23 rounds x to the nearest-even integer (kf = s - 1.5*2^23), r = x - kf in [-0.5, 0.5],
 * 2^r by a degree-5 polynomial (minimax on that interval, <= 2.8 ulp overall, exp2(0) == 1, never above 1),
 * scaled by 2^kf through the exponent field.                                                              */
float gso_exp2f(float x)
{
    const float magic = 12582912.0f;
    float s = x + magic;
    float kf = s - magic;
    float r = x - kf;
    float p = 1.3292919611558318e-3f;
    p = fmaf(p, r, 9.671509265899658e-3f);
    p = fmaf(p, r, 5.550636723637581e-2f);
    p = fmaf(p, r, 2.4022242426872253e-1f);
    p = fmaf(p, r, 6.931470632553101e-1f);
    float y = fmaf(p, r, 1.0f);
    int32_t k = (int32_t)kf;
    uint32_t bits;
    memcpy(&bits, &y, 4);
    bits += (uint32_t)k << 23;
    memcpy(&y, &bits, 4);
    return y;
}

/* Contract v3: the fragment's alpha is formed in the LOG domain, alpha = 2^(la - |kappa q|^2) with la = log2(opacity) formed
 * once per splat -- on the GPU that is one subtraction and one transcendental-unit instruction instead of a thirteen-operation
 * software 2^x and a multiply -- and the discard test (alpha < 1/255, shaders/GSplatShaderSource.h:307-308) is taken on the
 * ARGUMENT, la - pw < -log2(255), which both sides compute with the same float32 operations: oracle and product agree on
 * every fragment's fate, and 2^x itself may be any implementation good to a few ulp (here gso_exp2f, there v_exp_f32).
 * la: -inf unless opacity >= 1/255 (alpha <= opacity: such a splat can never pass the test; NaN lands here too), +inf for
 * +inf, else e + log2(m), opacity = m * 2^e with m in (sqrt(1/2), sqrt(2)], log2(m) = 2/ln2 * atanh(s), s = (m-1)/(m+1),
 * atanh by its series up to s^9 (|s| <= 0.1716: the first dropped term is < 2e-9 of the result).                        */
float gso_log2_opacity(float opacity)
{
    if (!(opacity >= 1.0f / 255.0f)) return -INFINITY;
    if (opacity > 3.4028234663852886e38f) return INFINITY;
    uint32_t bits;
    memcpy(&bits, &opacity, 4);
    int32_t e = (int32_t)(bits >> 23) - 127;
    bits = (bits & 0x007fffffu) | 0x3f800000u;
    float m;
    memcpy(&m, &bits, 4);                      /* [1, 2) */
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    float p = 1.0f / 9.0f;
    p = fmaf(p, z, 1.0f / 7.0f);
    p = fmaf(p, z, 1.0f / 5.0f);
    p = fmaf(p, z, 1.0f / 3.0f);
    p = fmaf(p, z, 1.0f);
    const float l = s * p;
    return fmaf(l, 2.885390043f, (float)e);   /* 2 / ln 2 */
}

/* src/GSplatRenderer.C:155-163 (texture side length); kept as a KAT target */
unsigned gso_closest_sqrt_power_of_2(int n)
{
    if (n <= 1) return 2;
    float s = sqrtf((float)n);
    unsigned p = (unsigned)ceilf(log2f(s));
    return 1u << p;
}

/* ------------------------------------------------------------------------- */
/* affine 4x3 row: m(r,0)*x + m(r,1)*y + m(r,2)*z + m(r,3) as an fmaf chain   */
static inline float aff(const float* m, int r, float x, float y, float z)
{
    return fmaf(M4(m, r, 0), x, fmaf(M4(m, r, 1), y, fmaf(M4(m, r, 2), z, M4(m, r, 3))));
}
/* linear 3x3 row */
static inline float lin(const float* m, int r, float x, float y, float z)
{
    return fmaf(M4(m, r, 0), x, fmaf(M4(m, r, 1), y, M4(m, r, 2) * z));
}

/* SH constants: shaders/GSplatShaderCoreLib.h:103-115 */
static const float SH_C1 = 0.4886025f;
static const float SH_C2_0 = 1.0925484f, SH_C2_1 = -1.0925484f, SH_C2_2 = 0.3153916f,
                   SH_C2_3 = -1.0925484f, SH_C2_4 = 0.5462742f;
static const float SH_C3_0 = -0.5900436f, SH_C3_1 = 2.8906114f, SH_C3_2 = -0.4570458f,
                   SH_C3_3 = 0.3731763f, SH_C3_4 = -0.4570458f, SH_C3_5 = 1.4453057f,
                   SH_C3_6 = -0.5900436f;

/* ShadeSH for one colour channel (shaders/GSplatShaderCoreLib.h:117-179).
 * sh[j] = coefficient sh(j+1) of this channel.  Expressions are evaluated
 * left to right exactly as written there, without fusing.                   */
static float gso_finite_colour(float c)
{
    if (c != c) return 0.0f;
    return c > 3.0e38f ? 3.0e38f : (c < -3.0e38f ? -3.0e38f : c);
}

static float shade_sh_channel(float base, const float* sh, float x, float y, float z, int order)
{
    float res = base;
    if (order >= 1) {
        res += SH_C1 * (-sh[0] * y + sh[1] * z - sh[2] * x);
        if (order >= 2) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            res += (SH_C2_0 * xy) * sh[3] + (SH_C2_1 * yz) * sh[4] +
                   (SH_C2_2 * (2.0f * zz - xx - yy)) * sh[5] + (SH_C2_3 * xz) * sh[6] +
                   (SH_C2_4 * (xx - yy)) * sh[7];
            if (order >= 3) {
                res += (SH_C3_0 * y * (3.0f * xx - yy)) * sh[8] + (SH_C3_1 * xy * z) * sh[9] +
                       (SH_C3_2 * y * (4.0f * zz - xx - yy)) * sh[10] +
                       (SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * sh[11] +
                       (SH_C3_4 * x * (4.0f * zz - xx - yy)) * sh[12] +
                       (SH_C3_5 * z * (xx - yy)) * sh[13] + (SH_C3_6 * x * (xx - 3.0f * yy)) * sh[14];
            }
        }
    }
    return fmaxf(res, 0.0f); /* max(res, vec3(0)) :178 */
}

/* Covariance chain shared by the beauty path and the wireframe overlay.  use_object = 0 for the wire
 * program, which never multiplies by transpose(mat3(glH_ObjectMatrix)) (shaders/GSplatShaderSource.h:75-77). */
#define OB(r, c) (use_object ? M4(f->object, (r), (c)) : ((r) == (c) ? 1.0f : 0.0f))
/* Returns 0 when the projected covariance is not finite (inf/NaN attribute halves, overflow): the contract drops such a
 * splat (GLSL leaves min(NaN, 4096) undefined; it must not become a screen-filling quad).                         */
static int covariance_axes(const gso_frame* f, int use_object, float x, float y, float z, float sx, float sy, float sz,
                            float qi, float qj, float qk, float qr, float* pex, float* pey, float* ps1, float* ps2)
{
    const float W = (float)f->width;
    /* CalcMatrixFromRotationScale (CoreLib :10-27): ms*mr with mr's COLUMNS
     * being the rows of the standard rotation matrix R(q); no normalisation */
    float R[3][3];
    R[0][0] = 1.0f - 2.0f * fmaf(qj, qj, qk * qk);
    R[0][1] = 2.0f * fmaf(qi, qj, -(qr * qk));
    R[0][2] = 2.0f * fmaf(qi, qk, qr * qj);
    R[1][0] = 2.0f * fmaf(qi, qj, qr * qk);
    R[1][1] = 1.0f - 2.0f * fmaf(qi, qi, qk * qk);
    R[1][2] = 2.0f * fmaf(qj, qk, -(qr * qi));
    R[2][0] = 2.0f * fmaf(qi, qk, -(qr * qj));
    R[2][1] = 2.0f * fmaf(qj, qk, qr * qi);
    R[2][2] = 1.0f - 2.0f * fmaf(qi, qi, qj * qj);
    /* M = diag(s) * R^T ; then * transpose(mat3(glH_ObjectMatrix)) (:231) */
    const float sc[3] = {sx, sy, sz};
    float M0[3][3], Mm[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) M0[a][b] = sc[a] * R[b][a];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            Mm[a][b] = fmaf(M0[a][2], OB(b, 2),
                            fmaf(M0[a][1], OB(b, 1), M0[a][0] * OB(b, 0)));
    /* CalcCovariance3D (CoreLib :29-35): sigma = transpose(M) * M */
    float S[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = a; b < 3; ++b) {
            float v = fmaf(Mm[2][a], Mm[2][b], fmaf(Mm[1][a], Mm[1][b], Mm[0][a] * Mm[0][b]));
            S[a][b] = v;
            S[b][a] = v;
        }

    /* CalcCovariance2D (CoreLib :38-76).  Note the VIEW matrix (no object
     * matrix) is used here (shader :240). */
    float tx = aff(f->view, 0, x, y, z);
    float ty = aff(f->view, 1, x, y, z);
    const float tz = aff(f->view, 2, x, y, z);
    {
        const float p00 = M4(f->proj, 0, 0), p11 = M4(f->proj, 1, 1);
        const float aspect = p00 / p11;
        const float tanFovX = 1.0f / p00;
        const float tanFovY = 1.0f / (p11 * aspect); /* == tanFovX up to rounding */
        const float limX = 1.3f * tanFovX, limY = 1.3f * tanFovY;
        float rx = tx / tz, ry = ty / tz;
        rx = fminf(fmaxf(rx, -limX), limX);
        ry = fminf(fmaxf(ry, -limY), limY);
        tx = rx * tz;
        ty = ry * tz;
    }
    const float focal = (W * M4(f->proj, 0, 0)) * 0.5f;
    const float j00 = focal / tz;
    const float tz2 = tz * tz;
    const float j02 = -(focal * tx) / tz2;
    const float j12 = -(focal * ty) / tz2;
    /* rows of A = J * mat3(view) (2x3); cov = A * sigma * A^T */
    float A0[3], A1[3];
    for (int c = 0; c < 3; ++c) {
        A0[c] = fmaf(j00, M4(f->view, 0, c), j02 * M4(f->view, 2, c));
        A1[c] = fmaf(j00, M4(f->view, 1, c), j12 * M4(f->view, 2, c));
    }
    float u0[3], u1[3];
    for (int k = 0; k < 3; ++k) {
        u0[k] = fmaf(S[k][2], A0[2], fmaf(S[k][1], A0[1], S[k][0] * A0[0]));
        u1[k] = fmaf(S[k][2], A1[2], fmaf(S[k][1], A1[1], S[k][0] * A1[0]));
    }
    const float cov00 = fmaf(A0[2], u0[2], fmaf(A0[1], u0[1], A0[0] * u0[0]));
    const float cov01 = fmaf(A0[2], u1[2], fmaf(A0[1], u1[1], A0[0] * u1[0]));
    const float cov11 = fmaf(A1[2], u1[2], fmaf(A1[1], u1[1], A1[0] * u1[0]));
    const float ca = cov00 + 0.3f; /* low-pass :72-74 */
    const float cb = cov01;
    const float cc = cov11 + 0.3f;

    /* DecomposeCovariance (CoreLib :79-93) */
    const float mid = 0.5f * (ca + cc);
    const float hd = (ca - cc) * 0.5f;
    const float radius = sqrtf(fmaf(hd, hd, cb * cb));
    const float lambda1 = mid + radius;
    const float lambda2 = fmaxf(mid - radius, 0.1f);
    const float dvx = cb, dvy = lambda1 - ca;
    const float dlen = sqrtf(fmaf(dvx, dvx, dvy * dvy));
    if (dlen > 0.0f) {
        *pex = dvx / dlen;
        *pey = dvy / dlen;
    } else { /* normalize(vec2(0)) is undefined in GLSL: choose the
                mathematically right eigenvector of a diagonal matrix with a>=c */
        *pex = 1.0f;
        *pey = 0.0f;
    }
    /* The shader negates diagVec.y (:89) and later out_vertex.y (:281); the
     * two flips cancel in GL window coordinates, leaving axes s1*e, s2*e_perp. */
    *ps1 = fminf(sqrtf(2.0f * lambda1), 4096.0f);
    *ps2 = fminf(sqrtf(2.0f * lambda2), 4096.0f);
    return fabsf(lambda1) < 3.0e38f;
}
#undef OB

/* Vertex stage for one splat.  Mirrors main() of the main vertex shader
 * (shaders/GSplatShaderSource.h:190-288) evaluated once per splat instead of
 * once per quad corner; the corner placement (:276-282) becomes the analytic
 * quad {c + qx*s1*e + qy*s2*e_perp, |qx|,|qy| <= 2} in GL window coordinates. */
static void project_splat(const gso_splats* s, const gso_frame* f, int64_t i, gso_record* o)
{
    memset(o, 0, sizeof(*o));
    const float W = (float)f->width, H = (float)f->height;

    const float px = s->P[3 * i + 0], py = s->P[3 * i + 1], pz = s->P[3 * i + 2];

    /* sort key: squared distance of the UN-offset point to the camera
     * (src/GSplatRenderer.C:197-201, mySplatPoints :454) */
    {
        float dx = px - f->cam_pos[0], dy = py - f->cam_pos[1], dz = pz - f->cam_pos[2];
        o->key = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    }

    /* texel 0 holds fl32(P - origin) (src/GSplatRenderer.C:459-461); the
     * shader adds the origin back (shader :201-202) */
    const float x = (px - f->origin[0]) + f->origin[0];
    const float y = (py - f->origin[1]) + f->origin[1];
    const float z = (pz - f->origin[2]) + f->origin[2];

    /* centerViewPos / centerClipPos with the flip-Y sandwich (:204-207) */
    const float tvx = aff(f->obj_view, 0, x, y, z);
    const float tvy = aff(f->obj_view, 1, x, y, z);
    const float tvz = aff(f->obj_view, 2, x, y, z);
    const float ftvy = -tvy;
    const float clx = aff(f->proj, 0, tvx, ftvy, tvz);
    const float cly = aff(f->proj, 1, tvx, ftvy, tvz);
    const float clz = aff(f->proj, 2, tvx, ftvy, tvz);
    const float clw = aff(f->proj, 3, tvx, ftvy, tvz);

    if (!(clw > 0.0f)) return; /* :209-214 behind camera */
    /* all four corners share the centre's z,w (:277-279): GL near/far clip
     * drops the whole quad iff z is outside [-w, w] */
    if (clz < -clw || clz > clw) return;

    /* out_vertex.y = -out_vertex.y (:281) undoes the flip for the centre */
    const float ndcx = clx / clw;
    const float ndcy = (-cly) / clw;
    o->cx = fmaf(ndcx, 0.5f, 0.5f) * W;
    o->cy = fmaf(ndcy, 0.5f, 0.5f) * H;
    /* every corner carries the centre's z and w: one window depth per quad (default depth range 0..1) */
    o->zwin = fmaf(clz / clw, 0.5f, 0.5f);

    /* attributes (:217-222); fp16 values are exact in fp32 */
    const float sx = gso_half_to_float(s->scale[3 * i + 0]);
    const float sy = gso_half_to_float(s->scale[3 * i + 1]);
    const float sz = gso_half_to_float(s->scale[3 * i + 2]);
    /* orient.wxyz -> rot (:230): rot.x = w (real), rot.yzw = xyz */
    const float qi = gso_half_to_float(s->orient[4 * i + 0]);
    const float qj = gso_half_to_float(s->orient[4 * i + 1]);
    const float qk = gso_half_to_float(s->orient[4 * i + 2]);
    const float qr = gso_half_to_float(s->orient[4 * i + 3]);

    float ex, ey, s1, s2;
    if (!covariance_axes(f, 1, x, y, z, sx, sy, sz, qi, qj, qk, qr, &ex, &ey, &s1, &s2)) return;
    o->ex = ex;
    o->ey = ey;
    o->is1 = 1.0f / s1;
    o->is2 = 1.0f / s2;
    /* contract v2: quad-local coordinate scaled by kappa = sqrt(log2 e), as two affine forms of the pixel position
     * (the fragment shader's interpolated fsIn.pos, shaders/GSplatShaderSource.h:276-282,305-306, times kappa):
     *   kq0 = d . (a1x, a1y),  kq1 = d . (b1x, b1y),  d = pixel centre - quad centre                         */
    {
        const float k1 = o->is1 * GSO_KAPPA, k2 = o->is2 * GSO_KAPPA;
        o->a1x = ex * k1;
        o->a1y = ey * k1;
        o->b1x = -(ey * k2);
        o->b1y = ex * k2;
    }
    /* conservative bbox of the +-2 quad (padding covers rounding in q) */
    o->hx = fmaf(2.0f * fmaf(s1, fabsf(ex), s2 * fabsf(ey)), 1.0001f, 0.01f);
    o->hy = fmaf(2.0f * fmaf(s1, fabsf(ey), s2 * fabsf(ex)), 1.0001f, 0.01f);

    /* colour (:224, :244-274) */
    float cr = gso_half_to_float(s->Cd[3 * i + 0]);
    float cg = gso_half_to_float(s->Cd[3 * i + 1]);
    float cbl = gso_half_to_float(s->Cd[3 * i + 2]);
    if (f->sh_order > 0 && s->shx) {
        float shr[15], shg[15], shb[15];
        for (int j = 0; j < 15; ++j) { /* coefficient j at (j/4, j%4) of the 4x4 (GR_GSplat.C:345-353) */
            shr[j] = gso_half_to_float(s->shx[16 * i + j]);
            shg[j] = gso_half_to_float(s->shy[16 * i + j]);
            shb[j] = gso_half_to_float(s->shz[16 * i + j]);
        }
        const float wx = x - f->cam_pos[0], wy = y - f->cam_pos[1], wz = z - f->cam_pos[2];
        const float ox = lin(f->inv_object, 0, wx, wy, wz);
        const float oy = lin(f->inv_object, 1, wx, wy, wz);
        const float oz = lin(f->inv_object, 2, wx, wy, wz);
        const float len = sqrtf(fmaf(oz, oz, fmaf(oy, oy, ox * ox)));
        const float dx = ox / len, dy = oy / len, dz = oz / len;
        cr = shade_sh_channel(cr, shr, dx, dy, dz, f->sh_order);
        cg = shade_sh_channel(cg, shg, dx, dy, dz, f->sh_order);
        cbl = shade_sh_channel(cbl, shb, dx, dy, dz, f->sh_order);
    }
    /* A colour that is not finite (inf / NaN colour or SH halves) would turn every pixel it is blended into -- and, in a
     * renderer that blends rejected fragments with weight 0, every pixel near it -- into NaN.  The contract (round 6) makes it
     * finite where it is formed: NaN -> 0 (what max(NaN, 0) gives in :178), +-inf -> +-3e38.  Nothing a real capture holds. */
    o->r = gso_finite_colour(cr);
    o->g = gso_finite_colour(cg);
    o->b = gso_finite_colour(cbl);
    o->opacity = s->alpha[i];
    o->la = gso_log2_opacity(o->opacity);
    o->visible = 1;
}

int gso_preprocess(const gso_splats* s, const gso_frame* f, gso_record* rec)
{
    if (!s || !f || !rec) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < s->n; ++i) project_splat(s, f, i, &rec[i]);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* stable LSD radix argsort on the IEEE bits of non-negative float keys.
 * The reference uses an unstable tbb::parallel_sort (src/GSplatRenderer.C:206)
 * whose tie order is unspecified; the contract fixes (key, index).          */
/* order0 (or NULL = index order): the order ties are left in -- the product stores the splats in an upload-time
 * order of its own (Morton order of the positions) and its stable sort breaks ties by that STORAGE order; a test
 * that compares depth orders index for index hands the storage order in.                                       */
static int argsort_keys_from(const float* keys, int64_t n, int32_t* perm, const int32_t* order0)
{
    uint32_t* k0 = (uint32_t*)malloc((size_t)n * 4 + 4);
    uint32_t* k1 = (uint32_t*)malloc((size_t)n * 4 + 4);
    int32_t* p1 = (int32_t*)malloc((size_t)n * 4 + 4);
    if (!k0 || !k1 || !p1) { free(k0); free(k1); free(p1); return -2; }
    for (int64_t i = 0; i < n; ++i) {
        uint32_t b;
        memcpy(&b, &keys[i], 4);
        /* keys are sums of squares: >= +0 or NaN; map to an order-preserving
         * uint (general form, handles -0 too) */
        b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        k1[i] = b;
    }
    for (int64_t j = 0; j < n; ++j) {
        const int32_t i = order0 ? order0[j] : (int32_t)j;
        if (i < 0 || (int64_t)i >= n) { free(k0); free(k1); free(p1); return -1; }
        k0[j] = k1[i];
        perm[j] = i;
    }
    uint32_t* ks = k0; uint32_t* kd = k1;
    int32_t* ps = perm; int32_t* pd = p1;
    for (int pass = 0; pass < 4; ++pass) {
        int64_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        const int sh = pass * 8;
        for (int64_t i = 0; i < n; ++i) ++cnt[((ks[i] >> sh) & 255u) + 1];
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < n; ++i) {
            int64_t dst = cnt[(ks[i] >> sh) & 255u]++;
            kd[dst] = ks[i];
            pd[dst] = ps[i];
        }
        uint32_t* tk = ks; ks = kd; kd = tk;
        int32_t* tp = ps; ps = pd; pd = tp;
    }
    /* 4 passes: result is back in (k0, perm) */
    free(k0); free(k1); free(p1);
    return 0;
}

int gso_argsort_from(const gso_record* rec, int64_t n, const int32_t* order0, int32_t* perm);
/* ------------------------------------------------------------------------- */
/* Tie order.  The reference's tbb::parallel_sort is unstable (src/GSplatRenderer.C:206-207): which of two splats at
 * exactly the same distance is drawn first is unspecified there.  The contract fixes it: equal keys are drawn in STORAGE
 * ORDER, and the storage order is the stable order of the 30-bit Morton codes of the positions, quantised to 10 bits per
 * axis inside the bounding box of the cloud (upload order if a position is not finite, or by request).  The product
 * stores its splats in that order (k_cluster.h); this is the independent restatement.                              */
static int g_tie_upload_order = 0;
void gso_set_tie_order(int upload_order) { g_tie_upload_order = upload_order ? 1 : 0; }

static uint32_t spread10(uint32_t v)
{
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

int gso_storage_order(const float* P, int64_t n, int32_t* order)
{
    if (!order || (n > 0 && !P)) return -1;
    for (int64_t i = 0; i < n; ++i) order[i] = (int32_t)i;
    if (g_tie_upload_order || n < 2) return 0;
    double lo[3] = {3.0e38, 3.0e38, 3.0e38}, hi[3] = {-3.0e38, -3.0e38, -3.0e38};
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const float v = P[3 * i + k];
            if (!(fabsf(v) < 3.0e38f)) return 0;          /* inf / NaN: upload order */
            if (v < lo[k]) lo[k] = v;
            if (v > hi[k]) hi[k] = v;
        }
    float flo[3], fsc[3];
    for (int k = 0; k < 3; ++k) {
        const double ext = hi[k] - lo[k];
        flo[k] = (float)lo[k];
        fsc[k] = ext > 0.0 ? (float)(1023.999 / ext) : 0.0f;
        if (!(fabsf(fsc[k]) < 3.0e38f)) fsc[k] = 0.0f;
    }
    uint32_t* code = (uint32_t*)malloc((size_t)n * 4 + 4);
    uint32_t* c1 = (uint32_t*)malloc((size_t)n * 4 + 4);
    int32_t* o1 = (int32_t*)malloc((size_t)n * 4 + 4);
    if (!code || !c1 || !o1) { free(code); free(c1); free(o1); return -2; }
    for (int64_t i = 0; i < n; ++i) {
        uint32_t q[3];
        for (int k = 0; k < 3; ++k) {
            float t = (P[3 * i + k] - flo[k]) * fsc[k];
            t = fminf(fmaxf(t, 0.0f), 1023.0f);
            q[k] = (uint32_t)t;
        }
        code[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    }
    /* stable LSD radix sort of (code, index), 4 x 8 bits */
    uint32_t* ks = code; uint32_t* kd = c1;
    int32_t* ps = order; int32_t* pd = o1;
    for (int pass = 0; pass < 4; ++pass) {
        int64_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        const int sh = pass * 8;
        for (int64_t i = 0; i < n; ++i) ++cnt[((ks[i] >> sh) & 255u) + 1];
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < n; ++i) {
            const int64_t dst = cnt[(ks[i] >> sh) & 255u]++;
            kd[dst] = ks[i];
            pd[dst] = ps[i];
        }
        uint32_t* tk = ks; ks = kd; kd = tk;
        int32_t* tp = ps; ps = pd; pd = tp;
    }
    free(code); free(c1); free(o1);
    return 0;
}

/* argsort of the records' keys, ties in the storage order of the positions P */
static int argsort_records(const gso_record* rec, const float* P, int64_t n, int32_t* perm)
{
    int32_t* order = (int32_t*)malloc((size_t)n * 4 + 4);
    if (!order) return -2;
    int rc = gso_storage_order(P, n, order);
    if (!rc) rc = gso_argsort_from(rec, n, order, perm);
    free(order);
    return rc;
}

int gso_argsort_from(const gso_record* rec, int64_t n, const int32_t* order0, int32_t* perm)
{
    float* keys = (float*)malloc((size_t)n * 4 + 4);
    if (!keys) return -2;
    for (int64_t i = 0; i < n; ++i) keys[i] = rec[i].key;
    int rc = argsort_keys_from(keys, n, perm, order0);
    free(keys);
    return rc;
}

int gso_argsort(const gso_record* rec, int64_t n, int32_t* perm) { return gso_argsort_from(rec, n, (const int32_t*)0, perm); }

int gso_host_sort_only(const float* P, int64_t n, const float cam_pos[3], int32_t* perm)
{
    int32_t* order = (int32_t*)malloc((size_t)n * 4 + 4);
    if (!order) return -2;
    int rc = gso_storage_order(P, n, order);
    if (!rc) rc = gso_host_sort_from(P, n, cam_pos, order, perm);
    free(order);
    return rc;
}

int gso_host_sort_from(const float* P, int64_t n, const float cam_pos[3], const int32_t* order0, int32_t* perm)
{
    float* keys = (float*)malloc((size_t)n * 4 + 4);
    if (!keys) return -2;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float dx = P[3 * i] - cam_pos[0], dy = P[3 * i + 1] - cam_pos[1], dz = P[3 * i + 2] - cam_pos[2];
        keys[i] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    }
    int rc = argsort_keys_from(keys, n, perm, order0);
    free(keys);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Fragment shader + blend for one splat over pixel rows [row_lo, row_hi].
 * FS: shaders/GSplatShaderSource.h:304-312.  Blend: src factor
 * ONE_MINUS_DST_ALPHA, dst factor ONE, equation ADD, for colour and alpha
 * (src/GSplatRenderer.C:613-621).                                           */
/* While a frame is being composited the ALPHA channel of rgba holds the transmittance T = 1 - A (cleared to 1 by
 * begin_frame); finish_frame turns it into A.  Carrying T saves the product one subtraction per fragment.        */
static void begin_frame(float* rgba, int width, int height)
{
    const size_t n = (size_t)width * (size_t)height;
    for (size_t p = 0; p < n; ++p) { rgba[4 * p] = rgba[4 * p + 1] = rgba[4 * p + 2] = 0.0f; rgba[4 * p + 3] = 1.0f; }
}
static void finish_frame(float* rgba, int width, int height)
{
    const size_t n = (size_t)width * (size_t)height;
    for (size_t p = 0; p < n; ++p) rgba[4 * p + 3] = 1.0f - rgba[4 * p + 3];
}

static void splat_rows_depth(const gso_record* o, int width, int height, int row_lo, int row_hi,
                             const float* depth, float* rgba);

static void splat_rows(const gso_record* o, int width, int height, int row_lo, int row_hi, float* rgba)
{
    splat_rows_depth(o, width, height, row_lo, row_hi, (const float*)0, rgba);
}

static void splat_rows_depth(const gso_record* o, int width, int height, int row_lo, int row_hi,
                             const float* depth, float* rgba)
{
    (void)height;
    const float xlo = o->cx - o->hx - 0.5f, xhi = o->cx + o->hx - 0.5f;
    const float ylo = o->cy - o->hy - 0.5f, yhi = o->cy + o->hy - 0.5f;
    if (!(xhi >= 0.0f && xlo <= (float)(width - 1))) return;
    if (!(yhi >= (float)row_lo && ylo <= (float)row_hi)) return;
    const int i0 = (int)ceilf(fmaxf(xlo, 0.0f));
    const int i1 = (int)floorf(fminf(xhi, (float)(width - 1)));
    const int j0 = (int)ceilf(fmaxf(ylo, (float)row_lo));
    const int j1 = (int)floorf(fminf(yhi, (float)row_hi));
    for (int j = j0; j <= j1; ++j) {
        float* row = rgba + (size_t)j * (size_t)width * 4;
        /* Contract v2: positions are taken relative to the pixel's 16x16 TILE origin (the product composites per
         * tile; relative coordinates keep every product small, and the per-(splat, tile) constants c0/c1 are shared
         * by the tile's 256 pixels):  kq = l . a + ((tile origin - centre) . a),  l = pixel index inside the tile */
        const int tj = j & ~(GSO_TILE - 1);
        const float d0y = ((float)tj + 0.5f) - o->cy;
        const float ly = (float)(j - tj);
        for (int i = i0; i <= i1; ++i) {
            const int ti = i & ~(GSO_TILE - 1);
            const float d0x = ((float)ti + 0.5f) - o->cx;
            const float lx = (float)(i - ti);
            const float c0 = fmaf(d0x, o->a1x, d0y * o->a1y);
            const float c1 = fmaf(d0x, o->b1x, d0y * o->b1y);
            /* kappa * fsIn.pos: the quad-local coordinate, |q| <= 2 <=> |kq| <= 2 kappa */
            const float q0 = fmaf(lx, o->a1x, fmaf(ly, o->a1y, c0));
            const float q1 = fmaf(lx, o->b1x, fmaf(ly, o->b1y, c1));
            if (!(fmaxf(fabsf(q0), fabsf(q1)) <= GSO_QLIM)) continue; /* outside the quad */
            const float pw = fmaf(q0, q0, q1 * q1);                   /* = log2(e) * |q|^2 */
            const float arg = o->la - pw;                             /* log2 of exp(-|q|^2) * opacity (:306-307) */
            if (!(arg >= -GSO_LOG2_255)) continue;                    /* alpha < 1/255: discard (:307-308) */
            const float alpha = arg >= 0.0f ? 1.0f : gso_exp2f(arg);  /* clamp(alpha, 0, 1) */
            /* depth test against the opaque pass (depth writes are off): GL_LEQUAL */
            if (depth && !(o->zwin <= depth[(size_t)j * (size_t)width + (size_t)i])) continue;
            float* px = row + (size_t)i * 4;
            /* src = (rgb*alpha, alpha), factors (1 - dst.a, 1): C += (1-A)*alpha*rgb, A += (1-A)*alpha, with the
             * weight w = T*alpha formed once and T = 1 - A carried instead of A */
            const float w = px[3] * alpha;
            px[0] = fmaf(w, o->r, px[0]);
            px[1] = fmaf(w, o->g, px[1]);
            px[2] = fmaf(w, o->b, px[2]);
            px[3] = px[3] - w;
        }
    }
}

int gso_blend_serial(const gso_record* rec, const int32_t* perm, int64_t n, int width, int height, float* rgba)
{
    if (!rec || !perm || !rgba || width <= 0 || height <= 0) return -1;
    begin_frame(rgba, width, height);
    for (int64_t r = 0; r < n; ++r) { /* instance order = sorted order, nearest first */
        const gso_record* o = &rec[perm[r]];
        if (!o->visible) continue;
        splat_rows(o, width, height, 0, height - 1, rgba);
    }
    finish_frame(rgba, width, height);
    return 0;
}

int gso_blend_serial_depth(const gso_record* rec, const int32_t* perm, int64_t n, int width, int height,
                           const float* depth, float* rgba)
{
    if (!rec || !perm || !rgba || width <= 0 || height <= 0) return -1;
    begin_frame(rgba, width, height);
    for (int64_t r = 0; r < n; ++r) {
        const gso_record* o = &rec[perm[r]];
        if (!o->visible) continue;
        splat_rows_depth(o, width, height, 0, height - 1, depth, rgba);
    }
    finish_frame(rgba, width, height);
    return 0;
}

int gso_render_depth(const gso_splats* s, const gso_frame* f, const float* depth, float* rgba)
{
    if (!s || !f || !rgba) return -1;
    gso_record* rec = (gso_record*)malloc((size_t)(s->n + 1) * sizeof(gso_record));
    int32_t* perm = (int32_t*)malloc((size_t)(s->n + 1) * 4);
    if (!rec || !perm) { free(rec); free(perm); return -2; }
    int rc = gso_preprocess(s, f, rec);
    if (!rc) rc = argsort_records(rec, s->P, s->n, perm);
    if (!rc) rc = gso_blend_serial_depth(rec, perm, s->n, f->width, f->height, depth, rgba);
    free(rec);
    free(perm);
    return rc;
}

#define GSO_STRIP 8 /* rows per strip for the parallel renderer */

static int blend_parallel_rows(const gso_record* rec, const int32_t* perm, int64_t n, int width, int height,
                               int row_lo, int row_hi, float* rgba, int threads);

int gso_blend_parallel(const gso_record* rec, const int32_t* perm, int64_t n, int width, int height,
                       float* rgba, int threads)
{
    return blend_parallel_rows(rec, perm, n, width, height, 0, height - 1, rgba, threads);
}

/* rows [row_lo, row_hi] of the frame only (the rest of rgba is left cleared): full-size scenes in affordable pieces */
static int blend_parallel_rows(const gso_record* rec, const int32_t* perm, int64_t n, int width, int height,
                               int row_lo, int row_hi, float* rgba, int threads)
{
    if (!rec || !perm || !rgba || width <= 0 || height <= 0) return -1;
    if (row_lo < 0) row_lo = 0;
    if (row_hi > height - 1) row_hi = height - 1;
    begin_frame(rgba, width, height);
    const int nstrips = (height + GSO_STRIP - 1) / GSO_STRIP;
    /* bin splat ranks to strips, in rank order (two passes) */
    int64_t* start = (int64_t*)calloc((size_t)nstrips + 1, sizeof(int64_t));
    if (!start) return -2;
    int32_t* slo = (int32_t*)malloc((size_t)n * 4 + 4);
    int32_t* shi = (int32_t*)malloc((size_t)n * 4 + 4);
    if (!slo || !shi) { free(start); free(slo); free(shi); return -2; }
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        const gso_record* o = &rec[perm[r]];
        slo[r] = 1; shi[r] = 0;
        if (!o->visible) continue;
        const float ylo = o->cy - o->hy - 0.5f, yhi = o->cy + o->hy - 0.5f;
        const float xlo = o->cx - o->hx - 0.5f, xhi = o->cx + o->hx - 0.5f;
        if (!(yhi >= 0.0f && ylo <= (float)(height - 1))) continue;
        if (!(xhi >= 0.0f && xlo <= (float)(width - 1))) continue;
        const int j0 = (int)ceilf(fmaxf(ylo, 0.0f));
        const int j1 = (int)floorf(fminf(yhi, (float)(height - 1)));
        if (j1 < j0) continue;
        slo[r] = j0 / GSO_STRIP;
        shi[r] = j1 / GSO_STRIP;
    }
    for (int64_t r = 0; r < n; ++r)
        for (int s = slo[r]; s <= shi[r]; ++s) ++start[s + 1];
    for (int s = 0; s < nstrips; ++s) start[s + 1] += start[s];
    int32_t* list = (int32_t*)malloc((size_t)start[nstrips] * 4 + 4);
    int64_t* cur = (int64_t*)malloc(((size_t)nstrips + 1) * sizeof(int64_t));
    if (!list || !cur) { free(start); free(slo); free(shi); free(list); free(cur); return -2; }
    memcpy(cur, start, ((size_t)nstrips + 1) * sizeof(int64_t));
    for (int64_t r = 0; r < n; ++r)
        for (int s = slo[r]; s <= shi[r]; ++s) list[cur[s]++] = perm[r];
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int s = 0; s < nstrips; ++s) {
        int lo = s * GSO_STRIP;
        int hi = (lo + GSO_STRIP - 1 < height - 1) ? lo + GSO_STRIP - 1 : height - 1;
        if (lo < row_lo) lo = row_lo;
        if (hi > row_hi) hi = row_hi;
        if (hi < lo) continue;
        for (int64_t k = start[s]; k < start[s + 1]; ++k)
            splat_rows(&rec[list[k]], width, height, lo, hi, rgba);
    }
    finish_frame(rgba, width, height);
    free(start); free(slo); free(shi); free(list); free(cur);
    return 0;
}

int gso_render_rows(const gso_splats* s, const gso_frame* f, int row_lo, int row_hi, float* rgba, int threads)
{
    if (!s || !f || !rgba) return -1;
    gso_record* rec = (gso_record*)malloc((size_t)(s->n + 1) * sizeof(gso_record));
    int32_t* perm = (int32_t*)malloc((size_t)(s->n + 1) * 4);
    if (!rec || !perm) { free(rec); free(perm); return -2; }
    int rc = gso_preprocess(s, f, rec);
    if (!rc) rc = argsort_records(rec, s->P, s->n, perm);
    if (!rc) rc = blend_parallel_rows(rec, perm, s->n, f->width, f->height, row_lo, row_hi, rgba, threads);
    free(rec);
    free(perm);
    return rc;
}

int gso_render(const gso_splats* s, const gso_frame* f, float* rgba, int threads)
{
    if (!s || !f || !rgba) return -1;
    gso_record* rec = (gso_record*)malloc((size_t)(s->n + 1) * sizeof(gso_record));
    int32_t* perm = (int32_t*)malloc((size_t)(s->n + 1) * 4);
    if (!rec || !perm) { free(rec); free(perm); return -2; }
    int rc = gso_preprocess(s, f, rec);
    if (!rc) rc = argsort_records(rec, s->P, s->n, perm);
    if (!rc) {
        if (threads > 1)
            rc = gso_blend_parallel(rec, perm, s->n, f->width, f->height, rgba, threads);
        else
            rc = gso_blend_serial(rec, perm, s->n, f->width, f->height, rgba);
    }
    free(rec);
    free(perm);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Wireframe overlay: shaders/GSplatShaderSource.h:22-110, geometry src/GR_GSplat.C:374-421.
 * Line rule = the contract's stand-in for GL's diamond-exit rule (see k_wire.h / DESIGN.md).          */
static void wire_edge(float x0, float y0, float x1, float y1, int width, int height, uint64_t frag, uint64_t* zbuf)
{
    const float dx = x1 - x0, dy = y1 - y0;
    if (!(fabsf(dx) < 3.0e38f) || !(fabsf(dy) < 3.0e38f)) return;
    const int xmajor = fabsf(dx) >= fabsf(dy);
    const float m0 = xmajor ? x0 : y0, m1 = xmajor ? x1 : y1, n0 = xmajor ? y0 : x0;
    const float dm = xmajor ? dx : dy, dn = xmajor ? dy : dx;
    if (dm == 0.0f) return;
    const float lo = fminf(m0, m1), hi = fmaxf(m0, m1);
    const int mmax = (xmajor ? width : height) - 1, nmax = (xmajor ? height : width) - 1;
    const float flo = ceilf(lo - 0.5f), fhi = ceilf(hi - 0.5f) - 1.0f;
    if (!(fhi >= 0.0f && flo <= (float)mmax)) return;
    const int i0 = (int)fmaxf(flo, 0.0f), i1 = (int)fminf(fhi, (float)mmax);
    for (int i = i0; i <= i1; ++i) {
        const float t = (((float)i + 0.5f) - m0) / dm;
        const float nv = floorf(fmaf(t, dn, n0));
        if (!(nv >= 0.0f && nv <= (float)nmax)) continue;
        const int j = (int)nv;
        const size_t pix = xmajor ? ((size_t)j * (size_t)width + (size_t)i) : ((size_t)i * (size_t)width + (size_t)j);
        if (frag < zbuf[pix]) zbuf[pix] = frag;
    }
}

int gso_render_wire(const gso_splats* s, const gso_frame* f, float* rgba)
{
    if (!s || !f || !rgba) return -1;
    const int width = f->width, height = f->height;
    const size_t npix = (size_t)width * (size_t)height;
    uint64_t* zbuf = (uint64_t*)malloc(npix * 8 + 8);
    if (!zbuf) return -2;
    memset(zbuf, 0xff, npix * 8);
    const float W = (float)width, H = (float)height;
    for (int64_t i = 0; i < s->n; ++i) {
        const float x = s->P[3 * i], y = s->P[3 * i + 1], z = s->P[3 * i + 2]; /* no origin offset here */
        const float tvx = aff(f->obj_view, 0, x, y, z), tvy = aff(f->obj_view, 1, x, y, z), tvz = aff(f->obj_view, 2, x, y, z);
        const float ftvy = -tvy;
        const float clx = aff(f->proj, 0, tvx, ftvy, tvz), cly = aff(f->proj, 1, tvx, ftvy, tvz);
        const float clz = aff(f->proj, 2, tvx, ftvy, tvz), clw = aff(f->proj, 3, tvx, ftvy, tvz);
        if (!(clw > 0.0f) || clz < -clw || clz > clw) continue;
        const float cx = fmaf(clx / clw, 0.5f, 0.5f) * W;
        const float cy = fmaf((-cly) / clw, 0.5f, 0.5f) * H;
        const float zw = fmaf(clz / clw, 0.5f, 0.5f);
        const float sx = gso_half_to_float(s->scale[3 * i]), sy = gso_half_to_float(s->scale[3 * i + 1]),
                    sz = gso_half_to_float(s->scale[3 * i + 2]);
        const float qi = gso_half_to_float(s->orient[4 * i]), qj = gso_half_to_float(s->orient[4 * i + 1]),
                    qk = gso_half_to_float(s->orient[4 * i + 2]), qr = gso_half_to_float(s->orient[4 * i + 3]);
        float ex, ey, s1, s2;
        if (!covariance_axes(f, 0, x, y, z, sx, sy, sz, qi, qj, qk, qr, &ex, &ey, &s1, &s2)) continue;
        const float ax = (2.0f * s1) * ex, ay = (2.0f * s1) * ey;
        const float bx = (2.0f * s2) * (-ey), by = (2.0f * s2) * ex;
        const float c0x = (cx - ax) - bx, c0y = (cy - ay) - by;
        const float c1x = (cx + ax) - bx, c1y = (cy + ay) - by;
        const float c2x = (cx + ax) + bx, c2y = (cy + ay) + by;
        const float c3x = (cx - ax) + bx, c3y = (cy - ay) + by;
        uint32_t zb;
        memcpy(&zb, &zw, 4);
        const uint64_t frag = ((uint64_t)zb << 32) | (uint64_t)i;
        wire_edge(c0x, c0y, c1x, c1y, width, height, frag, zbuf);
        wire_edge(c1x, c1y, c2x, c2y, width, height, frag, zbuf);
        wire_edge(c2x, c2y, c3x, c3y, width, height, frag, zbuf);
        wire_edge(c3x, c3y, c0x, c0y, width, height, frag, zbuf);
    }
    for (size_t p = 0; p < npix; ++p) {
        float* o = rgba + p * 4;
        o[0] = o[1] = o[2] = o[3] = 0.0f;
        if (zbuf[p] != 0xffffffffffffffffull) {
            const int64_t i = (int64_t)(zbuf[p] & 0xffffffffull);
            o[0] = gso_half_to_float(s->Cd[3 * i]);
            o[1] = gso_half_to_float(s->Cd[3 * i + 1]);
            o[2] = gso_half_to_float(s->Cd[3 * i + 2]);
            o[3] = 1.0f;
        }
    }
    free(zbuf);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Where may a rasteriser legitimately disagree with the analytic quad?  A fragment's fate is decided by thresholds: the
 * quad's edges (GL decides coverage of a pixel centre that sits ON an edge by its fixed-point rules, after snapping the
 * vertices to its sub-pixel grid), the 1/255 discard, and -- depth-tested frames -- zwin <= depth.  mask[p] = 1 iff for
 * SOME visible splat pixel p lies within delta_px of one of the quad's four edges (and within delta_px of the quad along
 * the other axis), or inside the quad with log2(alpha * 255) within eps_log2 (+ what a shift of the quad by delta_px does to
 * it) of 0, or inside with |zwin - depth| <= eps_depth.  Every pixel of a reference-GLSL image that differs from the oracle's by more than the 1e-3 budget must lie
 * in this mask (tests/helpers.py): an error anywhere else is an arithmetic difference, not a coverage rule.            */
int gso_edge_mask(const gso_record* rec, int64_t n, int width, int height, float delta_px, float eps_log2,
                  const float* depth, float eps_depth, uint8_t* mask)
{
    if (!rec || !mask || width <= 0 || height <= 0) return -1;
    memset(mask, 0, (size_t)width * (size_t)height);
    for (int64_t r = 0; r < n; ++r) {
        const gso_record* o = &rec[r];
        if (!o->visible) continue;
        const float pad = delta_px + 1.0f;
        const float xlo = o->cx - o->hx - 0.5f - pad, xhi = o->cx + o->hx - 0.5f + pad;
        const float ylo = o->cy - o->hy - 0.5f - pad, yhi = o->cy + o->hy - 0.5f + pad;
        if (!(xhi >= 0.0f && xlo <= (float)(width - 1) && yhi >= 0.0f && ylo <= (float)(height - 1))) continue;
        const int i0 = (int)ceilf(fmaxf(xlo, 0.0f)), i1 = (int)floorf(fminf(xhi, (float)(width - 1)));
        const int j0 = (int)ceilf(fmaxf(ylo, 0.0f)), j1 = (int)floorf(fminf(yhi, (float)(height - 1)));
        /* |grad kq0| = |a1| = kappa / s1 per pixel: a distance of delta_px to the edge |kq0| = QLIM is |a1| delta_px in kq0 */
        const double na = sqrt((double)o->a1x * o->a1x + (double)o->a1y * o->a1y), nb = sqrt((double)o->b1x * o->b1x + (double)o->b1y * o->b1y);
        const double ta = na * delta_px, tb = nb * delta_px;
        for (int j = j0; j <= j1; ++j)
            for (int i = i0; i <= i1; ++i) {
                const double dx = ((double)i + 0.5) - o->cx, dy = ((double)j + 0.5) - o->cy;
                const double q0 = fabs(dx * o->a1x + dy * o->a1y), q1 = fabs(dx * o->b1x + dy * o->b1y);
                const double e0 = q0 - (double)GSO_QLIM, e1 = q1 - (double)GSO_QLIM;
                int hit = (fabs(e0) <= ta && e1 <= tb) || (fabs(e1) <= tb && e0 <= ta);
                if (!hit && e0 <= 0.0 && e1 <= 0.0) {
                    /* the 1/255 discard: a shift of the quad by delta_px moves |kq|^2 by up to 2 (|kq0| |a1| + |kq1| |b1|) delta_px */
                    const double arg = (double)o->la - (q0 * q0 + q1 * q1);
                    if (fabs(arg + (double)GSO_LOG2_255) <= eps_log2 + 2.0 * (q0 * ta + q1 * tb)) hit = 1;
                    if (depth && fabs((double)o->zwin - (double)depth[(size_t)j * (size_t)width + (size_t)i]) <= eps_depth) hit = 1;
                }
                if (hit) mask[(size_t)j * (size_t)width + (size_t)i] = 1;
            }
    }
    return 0;
}

/* How strongly does a pixel react to a sub-pixel shift of the quads that cover it?  A GL rasteriser snaps every vertex to its
 * sub-pixel grid before it interpolates the quad-local coordinate, so each quad of a reference image sits up to a grid step
 * away from where the shader put it: alpha = 2^(la - |kq|^2) of a fragment then moves by a factor 2^(-d|kq|^2),
 * d|kq|^2 <= 2 (|kq0| |a1| + |kq1| |b1|) * shift.  sens[p] = sum over the fragments of pixel p, in depth order, of
 * T * alpha * (|kq0| |a1| + |kq1| |b1|)   [1 / pixel]:
 * a shift of every quad by `shift` pixels changes a channel of p by at most ~ 2 ln2 * shift * sens[p] * (colour range).
 * Small splats (an axis of half a pixel) make this exceed the 1e-3 budget far from any quad edge (tests/helpers.py).        */
int gso_snap_sensitivity(const gso_record* rec, const int32_t* perm, int64_t n, int width, int height, float* sens)
{
    if (!rec || !perm || !sens || width <= 0 || height <= 0) return -1;
    const size_t npx = (size_t)width * (size_t)height;
    float* T = (float*)malloc(npx * sizeof(float) + 4);
    if (!T) return -2;
    for (size_t p = 0; p < npx; ++p) { T[p] = 1.0f; sens[p] = 0.0f; }
    for (int64_t r = 0; r < n; ++r) {
        const gso_record* o = &rec[perm[r]];
        if (!o->visible) continue;
        const float xlo = o->cx - o->hx - 0.5f, xhi = o->cx + o->hx - 0.5f;
        const float ylo = o->cy - o->hy - 0.5f, yhi = o->cy + o->hy - 0.5f;
        if (!(xhi >= 0.0f && xlo <= (float)(width - 1) && yhi >= 0.0f && ylo <= (float)(height - 1))) continue;
        const int i0 = (int)ceilf(fmaxf(xlo, 0.0f)), i1 = (int)floorf(fminf(xhi, (float)(width - 1)));
        const int j0 = (int)ceilf(fmaxf(ylo, 0.0f)), j1 = (int)floorf(fminf(yhi, (float)(height - 1)));
        const float na = sqrtf(o->a1x * o->a1x + o->a1y * o->a1y), nb = sqrtf(o->b1x * o->b1x + o->b1y * o->b1y);
        for (int j = j0; j <= j1; ++j)
            for (int i = i0; i <= i1; ++i) {
                const float dx = ((float)i + 0.5f) - o->cx, dy = ((float)j + 0.5f) - o->cy;
                const float q0 = dx * o->a1x + dy * o->a1y, q1 = dx * o->b1x + dy * o->b1y;
                if (!(fmaxf(fabsf(q0), fabsf(q1)) <= GSO_QLIM)) continue;
                const float arg = o->la - (q0 * q0 + q1 * q1);
                if (!(arg >= -GSO_LOG2_255)) continue;
                const float alpha = arg >= 0.0f ? 1.0f : exp2f(arg);
                const size_t p = (size_t)j * (size_t)width + (size_t)i;
                const float w = T[p] * alpha;
                sens[p] += w * (fabsf(q0) * na + fabsf(q1) * nb);
                T[p] -= w;
            }
    }
    free(T);
    return 0;
}

int gso_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

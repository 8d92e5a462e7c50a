/*
 * gsplat_oracle.h -- CPU ORACLE for the GSplat per-frame render path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it.  The product path
 * (houdini-gsplat-renderer_amd/) never includes, links or calls anything here.
 *
 * It is a from-scratch plain-C restatement of what the reference computes per
 * frame (all paths relative to /root/reference/gsplat_plugin):
 *   - per-splat vertex-shader math      shaders/GSplatShaderSource.h:190-288
 *   - covariance helpers                shaders/GSplatShaderCoreLib.h:10-93
 *   - spherical harmonics               shaders/GSplatShaderCoreLib.h:103-179
 *   - fragment shader                   shaders/GSplatShaderSource.h:304-312
 *   - fixed-function "under" blend      src/GSplatRenderer.C:613-621
 *   - camera-distance argsort           src/GSplatRenderer.C:176-216
 *   - origin offset round trip          src/GSplatRenderer.C:456-461 + shader :201-202
 *   - fp16 quantisation of attributes   src/GR_GSplat.C:315-318,345-367
 *
 * Parity pin: the reference ships no tests or golden vectors.  The oracle is
 * pinned instead against the reference's own GLSL executed on a software
 * GLES3 rasteriser in the build container (tests/golden/make_goldens.py);
 * the resulting images are committed under tests/golden/.
 *
 * Arithmetic contract ("float32 op order"): every operation below is IEEE-754
 * binary32, round-to-nearest-even, in exactly the order written.  fmaf() is a
 * single-rounding fused multiply-add.  Build with -ffp-contract=off so the
 * compiler never fuses anything that is not an explicit fmaf().  DESIGN.md
 * restates the contract; the HIP kernels implement it independently.
 */
#ifndef GSPLAT_ORACLE_H
#define GSPLAT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSO_TILE  16                 /* the product's tile edge (include/gsplat_hip.h GSR_TILE): fragment positions
                                        are formed relative to the tile origin (contract v2)                     */
#define GSO_KAPPA 1.2011224087864498f /* sqrt(log2 e): exp(-|q|^2) == 2^(-|kappa q|^2)                            */
#define GSO_LOG2_255 7.99435343685886f /* a fragment is discarded iff la - |kappa q|^2 < -log2(255)  (contract v3)    */
#define GSO_QLIM  (2.0f * GSO_KAPPA)  /* the quad |q| <= 2 in kappa units                                         */

/* Per-frame uniforms, named after the GLSL uniforms they stand for
 * (shaders/GSplatShaderSource.h:119-133,153-159).  Matrices are 16 floats in
 * GL column-major order, m[c*4+r] = element (row r, column c) of the matrix
 * that multiplies COLUMN vectors -- byte-identical to how Houdini's
 * UT_Matrix4F (row-vector convention, row-major storage) lands in a GLSL
 * mat4 uniform. */
typedef struct gso_frame {
    float obj_view[16];   /* glH_ObjViewMatrix                      */
    float object[16];     /* glH_ObjectMatrix                       */
    float inv_object[16]; /* glH_InvObjectMatrix                    */
    float view[16];       /* glH_ViewMatrix                         */
    float proj[16];       /* glH_ProjectMatrix                      */
    float cam_pos[3];     /* WorldSpaceCameraPos; also the sort reference point
                             (src/GSplatRenderer.C:551-563,584)     */
    float origin[3];      /* GSplatOrigin (src/GSplatRenderer.C:403-418) */
    int32_t width;        /* glH_ScreenSize.x                       */
    int32_t height;       /* glH_ScreenSize.y                       */
    int32_t sh_order;     /* GSplatShOrder after the doSH gate
                             (src/GSplatRenderer.C:623,628): 0..3   */
} gso_frame;

/* What the vertex stage hands to the fragment stage, per splat. */
typedef struct gso_record {
    float cx, cy;       /* quad centre, GL window coords (y up, pixel centres at +0.5) */
    float ex, ey;       /* unit major axis e; minor axis is (-ey, ex)                  */
    float is1, is2;     /* 1/s1, 1/s2 with s = min(sqrt(2*lambda), 4096)              */
    float a1x, a1y;     /* kappa * e / s1        (contract v2: kappa = sqrt(log2 e))  */
    float b1x, b1y;     /* kappa * e_perp / s2                                        */
    float hx, hy;       /* conservative half extents of the quad's bbox (not parity-relevant) */
    float r, g, b;      /* colour after SH                                            */
    float opacity;
    float la;           /* log2(opacity) by gso_log2_opacity() (contract v3): alpha = 2^(la - |kappa q|^2)   */
    float key;          /* squared distance to cam_pos (sort key)                     */
    float zwin;         /* window-space depth of the whole quad: ndc.z*0.5+0.5        */
    int32_t visible;    /* 0 if culled (w<=0, z outside [-w,w])                        */
} gso_record;

/* Splat attribute arrays use the reference's registerUpdate() layout
 * (include/GSplatRenderer.h:34-47):
 *   P       float [3N]   UT_Vector3Array
 *   Cd      half  [3N]   UT_Vector3HArray
 *   alpha   float [N]    UT_FloatArray
 *   scale   half  [3N]   UT_Vector3HArray
 *   orient  half  [4N]   UT_Vector4HArray  (x,y,z,w)
 *   shx/shy/shz half [16N] MyUT_Matrix4HArray, row-major 4x4, coefficient j
 *           (1-based j+1 = sh1..sh15) at (row j/4, col j%4); NULL = no SH.
 * halves are raw IEEE binary16 bit patterns (uint16_t). */
typedef struct gso_splats {
    int64_t n;
    const float*    P;
    const uint16_t* Cd;
    const float*    alpha;
    const uint16_t* scale;
    const uint16_t* orient;
    const uint16_t* shx;
    const uint16_t* shy;
    const uint16_t* shz;
} gso_splats;

/* scalar helpers (exposed for known-answer tests) */
float    gso_half_to_float(uint16_t h);
uint16_t gso_float_to_half(float f);              /* round-to-nearest-even */
float    gso_expf(float x);                       /* contract v1 exp, x in [-80, 0] (kept as a known-answer target) */
float    gso_exp2f(float x);                      /* the oracle's 2^x, x in [-100, 0] (contract v3 admits any 2^x good to a few ulp) */
float    gso_log2_opacity(float opacity);         /* the contract's log2 of a splat's opacity (-inf below 1/255) */
unsigned gso_closest_sqrt_power_of_2(int n);      /* src/GSplatRenderer.C:155-163 */

/* vertex stage for all splats; rec[n] */
int gso_preprocess(const gso_splats* s, const gso_frame* f, gso_record* rec);

/* Tie order of the contract (the reference's own is unspecified: unstable tbb::parallel_sort, src/GSplatRenderer.C:206-207):
 * equal keys are drawn in STORAGE ORDER = the stable order of the 30-bit Morton codes of the positions (10 bits per axis
 * inside the cloud's bounding box); upload order when a position is not finite.  order[j] = index of the j-th splat.
 * gso_set_tie_order(1) makes the storage order the upload order (the product's GSR_OPT_STORAGE_ORDER = 0).  Every whole-
 * frame entry point below (gso_render*, gso_host_sort_only) sorts with it. */
int gso_storage_order(const float* P, int64_t n, int32_t* order);
void gso_set_tie_order(int upload_order);

/* stable ascending argsort of rec[i].key over ALL n splats; ties: lower index first (gso_argsort), or left in the order of
 * order0[], a permutation of 0..n-1 (gso_argsort_from; NULL = index order) */
int gso_argsort(const gso_record* rec, int64_t n, int32_t* perm);
int gso_argsort_from(const gso_record* rec, int64_t n, const int32_t* order0, int32_t* perm);

/* fragment + blend stage, literal serial form.  rgba = float[height*width*4],
 * premultiplied, row 0 = bottom row (GL window coords), cleared to 0 first. */
int gso_blend_serial(const gso_record* rec, const int32_t* perm, int64_t n,
                     int width, int height, float* rgba);

/* same pixels, bit-identical, strip-parallel with OpenMP (CPU baseline). */
int gso_blend_parallel(const gso_record* rec, const int32_t* perm, int64_t n,
                       int width, int height, float* rgba, int threads);

/* Depth-tested variant: the reference draws after the opaque pass with the depth test
 * on and depth writes off (src/GSplatRenderer.C:595-610).  depth = float[height*width] window depth
 * (0..1, row 0 = bottom) of what is already in the framebuffer, or NULL for no test; a splat
 * fragment survives iff its quad's depth <= depth[pixel] (GL_LEQUAL). */
int gso_blend_serial_depth(const gso_record* rec, const int32_t* perm, int64_t n,
                           int width, int height, const float* depth, float* rgba);
int gso_render_depth(const gso_splats* s, const gso_frame* f, const float* depth, float* rgba);

/* wireframe overlay: outlines of the +-2 quads, colour Cd, alpha 1, nearest line wins */
int gso_render_wire(const gso_splats* s, const gso_frame* f, float* rgba);

/* whole frame: preprocess + argsort + blend.  threads<=1 -> serial blend. */
int gso_render(const gso_splats* s, const gso_frame* f, float* rgba, int threads);

/* the same frame, rows [row_lo, row_hi] only (other rows stay 0): full-size configs checked in pieces */
int gso_render_rows(const gso_splats* s, const gso_frame* f, int row_lo, int row_hi, float* rgba, int threads);

/* only the reference's per-camera-move HOST work: distances + argsort
 * (src/GSplatRenderer.C:188-208).  perm[n] out.  Ties in storage order (gso_host_sort_only) / in the order of order0. */
int gso_host_sort_only(const float* P, int64_t n, const float cam_pos[3], int32_t* perm);
int gso_host_sort_from(const float* P, int64_t n, const float cam_pos[3], const int32_t* order0, int32_t* perm);

/* number of OpenMP threads the parallel entry points would use */
/* pixels where a rasteriser's coverage / discard / depth rule may decide differently from the analytic quad (see the .c file) */
int gso_edge_mask(const gso_record* rec, int64_t n, int width, int height, float delta_px, float eps_log2,
                  const float* depth, float eps_depth, uint8_t* mask);
/* per-pixel sensitivity of the frame to a sub-pixel shift of its quads (see the .c file) */
int gso_snap_sensitivity(const gso_record* rec, const int32_t* perm, int64_t n, int width, int height, float* sens);
int gso_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif

/*
 * GSplatPrim.h -- HDK-free mirror of the reference's viewport primitive GR_PrimGsplat, as far as the render path
 * needs it: the step BEFORE GSplatRenderer::registerUpdate.
 *
 * What it mirrors (/root/reference/gsplat_plugin/src/GR_GSplat.C):
 *   update()  :192-457  attribute discovery with the reference's precedence and defaults, fp32 -> fp16 quantisation,
 *                       the three spherical-harmonics naming schemes, registerUpdate(), the detail attributes
 *                       gsplat__explicit_camera_pos and gsplat__sh_order (invalid order -> 0)
 *   render()  :459-492  setRenderingEnabled(beauty mode), includeInRenderPass, explicit camera, SH order
 *   ~dtor     :63-70    flushEntriesForMatchingDetail
 * HDK attribute handles become plain float pointers (NULL = attribute absent).  The wireframe geometry of update()
 * :374-421 is not built here: the overlay is drawn by gsr_render_wire from the same device arrays.
 */
#ifndef GSPLAT_PRIM_MIRROR_H
#define GSPLAT_PRIM_MIRROR_H

#include <stdint.h>

#include "GSplatRenderer.h"

/* Point attributes of ONE GSplat primitive in vertex order, as float32 (Houdini's native storage), and the two detail
 * attributes.  count = gSplatPrim->getVertexCount(). */
typedef struct gsplat_attrs {
    int64_t count;
    const float* P;                 /* [3*count]  required                                                     */
    const float* Cd;                /* [3*count]  or NULL -> (0,0,0)      + "not found" error (:233-238,310)  */
    const float* opacity;           /* [count]    "opacity"                                                    */
    const float* Alpha;             /* [count]    "Alpha": WINS when present (GSOPs compatibility, :240-257);
                                                  both NULL -> 1          + error (:243-246,311)               */
    const float* scale;             /* [3*count]  or NULL -> (1,1,1)      + error (:259-265,312)               */
    const float* orient;            /* [4*count]  x,y,z,w; NULL -> (0,0,0,1) + error (:267-272,313)            */
    /* spherical harmonics, tried in this order (:145-189): */
    const float* sh_coefficients;   /* float-array attribute "sh_coefficients": per point sh_coefficients_len floats */
    int32_t sh_coefficients_len;    /*   = 3 * (vec3 per point), at most 45; coefficient j -> (row j/4, col j%4)     */
    const float* const* sh;         /* NULL or 15 pointers "sh1".."sh15", each [3*count]; the scheme is used when sh[0]
                                       exists; attributes after the first missing one read as 0                       */
    const float* const* f_rest;     /* NULL or 45 pointers "f_rest_0".."f_rest_44", each [count]; used when f_rest[0]
                                       exists; coefficient j = (f_rest_j, f_rest_{j+15}, f_rest_{j+30}) (:357-367)    */
    const int32_t* sh_order;        /* detail attribute gsplat__sh_order or NULL (default 3, :444-457)                */
    const float* explicit_camera_pos; /* detail attribute gsplat__explicit_camera_pos [3] or NULL (:274-279,438-442) */
} gsplat_attrs;

/* which attributes update() had to default (bit mask returned by gsplat_prim_missing) */
#define GSPLAT_MISSING_CD      1
#define GSPLAT_MISSING_OPACITY 2
#define GSPLAT_MISSING_SCALE   4
#define GSPLAT_MISSING_ORIENT  8
#define GSPLAT_MISSING_SH      16   /* none of the three SH schemes: order-0 rendering (warning, :180-183) */
#define GSPLAT_BAD_SH_ORDER    32   /* gsplat__sh_order outside 0..3: contribution disabled (:447-451) */

#ifdef __cplusplus
#include <string>
#include <vector>

class GSplatPrim {
public:
    explicit GSplatPrim(GSplatRenderer& renderer) : renderer_(renderer) {}
    ~GSplatPrim();
    GSplatPrim(const GSplatPrim&) = delete;
    GSplatPrim& operator=(const GSplatPrim&) = delete;

    /* GR_PrimGsplat::update: ingest + registerUpdate.  detail/version/vtxOffset identify the primitive as in
     * registerUpdate(); barycenter = gSplatPrim->baryCenter() or NULL to take the float32 mean of P. */
    void update(const void* detail, const GSplatCacheVersion& version, int64_t vtxOffset, const gsplat_attrs& a,
                const float* barycenter);
    /* GR_PrimGsplat::render; beautyMode = (render_mode < GR_RENDER_NUM_BEAUTY_MODES) */
    void render(bool beautyMode);

    const std::string& registryId() const { return id_; }
    int shOrder() const { return sh_order_; }
    unsigned missing() const { return missing_; }
    int64_t count() const { return count_; }
    bool hasSh() const { return !shx_.empty(); }
    /* the quantised arrays the renderer borrows (registerUpdate layout) */
    const float* P() const { return P_.data(); }
    const uint16_t* Cd() const { return Cd_.data(); }
    const float* alpha() const { return alpha_.data(); }
    const uint16_t* scale() const { return scale_.data(); }
    const uint16_t* orient() const { return orient_.data(); }
    const uint16_t* shx() const { return shx_.data(); }
    const uint16_t* shy() const { return shy_.data(); }
    const uint16_t* shz() const { return shz_.data(); }

private:
    GSplatRenderer& renderer_;
    std::string id_;
    int64_t count_ = 0;
    std::vector<float> P_, alpha_;
    std::vector<uint16_t> Cd_, scale_, orient_, shx_, shy_, shz_;
    int sh_order_ = 3;
    bool has_eye_ = false;
    float eye_[3] = {0, 0, 0};
    unsigned missing_ = 0;
};

extern "C" {
#endif

typedef struct gsplat_prim gsplat_prim;
gsplat_prim* gsplat_prim_create(gsplat_renderer* renderer);
void gsplat_prim_destroy(gsplat_prim* p);                              /* flushes its registry entries */
/* returns the registry id length (id copied to id_out) or <0 */
int  gsplat_prim_update(gsplat_prim* p, uint64_t detail, const int64_t version[4], int64_t vtx_offset,
                        const gsplat_attrs* attrs, const float* barycenter_or_null, char* id_out, int id_cap);
void gsplat_prim_render(gsplat_prim* p, int beauty_mode);
unsigned gsplat_prim_missing(gsplat_prim* p);
int  gsplat_prim_sh_order(gsplat_prim* p);
int  gsplat_prim_has_sh(gsplat_prim* p);
/* what == 0..7: P, Cd, alpha, scale, orient, shx, shy, shz (NULL when absent) */
const void* gsplat_prim_array(gsplat_prim* p, int what);

#ifdef __cplusplus
}
#endif
#endif

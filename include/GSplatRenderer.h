/*
 * GSplatRenderer.h -- HDK-free host mirror of the reference's renderer core.
 *
 * Same class name, same nine verbs, same argument meaning and the same
 * silent-early-return error behaviour as
 * /root/reference/gsplat_plugin/include/GSplatRenderer.h:22-131, so that the
 * reference's callers (GR_PrimGsplat::update/render, src/GR_GSplat.C:423-436,
 * 472-492, and MyCustomSceneRenderHook::render, src/DM_GSplatHook.C:30-39)
 * keep their shape.  HDK types are replaced by PODs:
 *   GU_Detail*            -> const void* (identity only)
 *   RE_CacheVersion       -> GSplatCacheVersion (4 x int64)
 *   UT_Vector3Array & co. -> borrowed raw pointers + element count
 *   RE_RenderContext      -> GSplatRenderContext (the matrices/size the
 *                            reference pulls from `r`, plus the RGBA target)
 * The GL half (textures, shader, drawInstanced) is replaced by libgsplat_hip
 * (include/gsplat_hip.h).  See INTEGRATION.md for the HDK-side glue.
 */
#ifndef GSPLAT_RENDERER_MIRROR_H
#define GSPLAT_RENDERER_MIRROR_H

#include <stdint.h>

#include "gsplat_hip.h"

#ifdef __cplusplus
#include <map>
#include <memory>
#include <string>
#include <vector>

struct GSplatCacheVersion {
    int64_t e[4] = {0, 0, 0, 0};
    bool operator!=(const GSplatCacheVersion& o) const
    {
        return e[0] != o.e[0] || e[1] != o.e[1] || e[2] != o.e[2] || e[3] != o.e[3];
    }
};
#endif

/* What render()/generateRenderGeometry() need from the host's render context. */
typedef struct GSplatRenderContext {
    float obj_view[16];   /* glH_ObjViewMatrix   (GL column-major = UT_Matrix4F::data()) */
    float object[16];     /* glH_ObjectMatrix    */
    float inv_object[16]; /* glH_InvObjectMatrix */
    float view[16];       /* glH_ViewMatrix; also r->getMatrix() at hook time, whose inverse
                             gives the camera position (src/GSplatRenderer.C:558-562) */
    float proj[16];       /* glH_ProjectMatrix   */
    int32_t width, height;/* glH_ScreenSize      */
    float* target;        /* RGBA-f32 premultiplied, row 0 = bottom; height*width*4 floats */
    int32_t target_is_device;
    const float* depth;   /* optional: window depth (0..1, row 0 = bottom) left by the opaque pass; the splats are
                             depth-tested against it with depth writes off (src/GSplatRenderer.C:595-610); NULL = none */
    int32_t depth_is_device;
} GSplatRenderContext;

#ifdef __cplusplus

class GSplatRenderer {
public:
    static const int64_t GSPLAT_COUNT_MAX = 1 << 23; /* include/GSplatRenderer.h:26 */

    /* process-wide instance on GPU 0, as in the reference (:29-32) */
    static GSplatRenderer& getInstance();

    /* device >= 0: bind a libgsplat_hip context on that GPU;
     * device <  0: "dry" instance -- registry/staging logic only, no GPU (tests). */
    explicit GSplatRenderer(int device);
    /* several GPUs behind the same nine verbs, driven from the one draw thread (gsr_multi_*, gsplat_hip.h):
     * tile rows are sharded over `devices`, the frame lands on devices[0].  transport: GSR_TRANSPORT_*. */
    GSplatRenderer(const int* devices, int count, int transport);
    ~GSplatRenderer();
    GSplatRenderer(const GSplatRenderer&) = delete;
    GSplatRenderer& operator=(const GSplatRenderer&) = delete;

    /* Arrays are BORROWED until the next generateRenderGeometry() that stages
     * them (the reference keeps raw pointers too, src/GSplatRenderer.C:277-284).
     * shCount = element count of the three SH arrays: splatCount or 0. */
    std::string registerUpdate(const void* gdp, const GSplatCacheVersion& gversion, int64_t gVtxOffset,
                               int64_t splatCount, const float splatOrigin[3], const float* splatPts,
                               const uint16_t* splatColors, const float* splatAlphas,
                               const uint16_t* splatScales, const uint16_t* splatOrients,
                               const uint16_t* splatShxs, const uint16_t* splatShys,
                               const uint16_t* splatShzs, int64_t shCount);
    void includeInRenderPass(const std::string& gSplatId);
    void flushEntriesForMatchingDetail(const std::string& myRegistryId);
    void generateRenderGeometry(GSplatRenderContext& r);
    void render(GSplatRenderContext& r, bool isObjectLevel);
    void postRender();
    void setRenderingEnabled(bool isRenderEnabled);
    void setExplicitCameraPos(const float explicitCameraPos[3]);
    void setSphericalHarmonicsOrder(int shOrder);

    /* introspection (no reference counterpart; used by tests and the C wrappers) */
    enum Query {
        Q_REGISTRY_SIZE = 0, Q_ACTIVE_STAGED = 1, Q_SPLAT_COUNT = 2, Q_CAN_RENDER = 3,
        Q_STAGING_COUNT = 4,   /* how many times geometry was (re)staged */
        Q_RENDER_COUNT = 5,    /* how many frames reached the device */
        Q_SH_PRESENT = 6, Q_LAST_STATUS = 7, Q_ENTRY_AGE = 8, Q_ENTRY_AGE_SINCE_ACTIVE = 9
    };
    int64_t query(int what, const std::string& id = std::string()) const;
    void origin(float out[3]) const;
    void lastCameraPos(float out[3]) const;
    gsr_context* engine() const { return engine_; }
    gsr_multi* multi() const { return multi_; }

    static unsigned int closestSqrtPowerOf2(int n); /* src/GSplatRenderer.C:155-163 */

private:
    /* one registered primitive (what registerUpdate() was told; arrays borrowed) */
    struct Row {
        const void* detail = nullptr;
        GSplatCacheVersion version;
        int64_t count = 0;
        float origin[3] = {0, 0, 0};
        const float* P = nullptr;
        const uint16_t* Cd = nullptr;
        const float* alpha = nullptr;
        const uint16_t* scale = nullptr;
        const uint16_t* orient = nullptr;
        const uint16_t *shx = nullptr, *shy = nullptr, *shz = nullptr;
        int64_t sh_count = 0;
        bool shown = false;             /* includeInRenderPass() since the last postRender() */
        int redraws = -1;               /* postRender() calls since registration, minus one */
        int redraws_since_shown = -1;   /* -1 = never shown */
    };
    /* what one redraw puts on the GPU */
    struct Plan {
        struct Part { std::string id; int64_t take; };
        std::vector<Part> parts;        /* rows in packing order, with the number of splats taken from each */
        int64_t total = 0;              /* splats in the pass (<= 2^23 - 1) */
        int64_t wanted = 0;             /* splats the joined rows hold before truncation */
        int64_t registered = 0;         /* splats in the whole table */
        bool sh = false;
        float origin[3] = {0, 0, 0};
        bool sameAs(const Plan& o) const
        {
            if (parts.size() != o.parts.size() || total != o.total) return false;
            for (size_t k = 0; k < parts.size(); ++k)
                if (parts[k].id != o.parts[k].id || parts[k].take != o.parts[k].take) return false;
            return true;
        }
    };
    Plan planFrame() const;
    bool upload(const Plan& p);

    std::map<std::string, Row> table_;
    Plan resident_;                     /* the plan whose splats are in HBM */
    bool resident_ok_ = false;
    gsr_context* engine_ = nullptr;     /* one GPU ... */
    gsr_multi* multi_ = nullptr;        /* ... or several */
    bool dry_ = false;
    bool enabled_ = true;
    bool can_render_ = false;
    bool eye_override_ = false;
    float eye_explicit_[3] = {0, 0, 0};
    float eye_[3] = {0, 0, 0};
    int sh_order_ = 3;
    int64_t stagings_ = 0, frames_ = 0;
    int status_ = 0;
    bool obj_notice_given_ = false;
    bool greeted_ = false;
};

extern "C" {
#endif /* __cplusplus */

/* ---- flat C wrappers (ctypes / FFI) --------------------------------------- */
typedef struct gsplat_renderer gsplat_renderer;

gsplat_renderer* gsplat_renderer_create(int device);       /* device < 0: dry instance; NULL on failure */
gsplat_renderer* gsplat_renderer_create_multi(const int* devices, int count, int transport);   /* gsr_multi_create behind the verbs */
gsplat_renderer* gsplat_renderer_get_instance(void);       /* the singleton (GPU 0) */
void gsplat_renderer_destroy(gsplat_renderer* h);           /* not for the singleton */
/* writes the registry id (NUL-terminated) into id_out; returns its length or <0 */
int  gsplat_renderer_register_update(gsplat_renderer* h, uint64_t gdp, const int64_t gversion[4],
                                     int64_t gvtx_offset, int64_t splat_count, const float origin[3],
                                     const float* P, const uint16_t* Cd, const float* alpha,
                                     const uint16_t* scale, const uint16_t* orient,
                                     const uint16_t* shx, const uint16_t* shy, const uint16_t* shz,
                                     int64_t sh_count, char* id_out, int id_cap);
void gsplat_renderer_include_in_render_pass(gsplat_renderer* h, const char* id);
void gsplat_renderer_flush_entries_for_matching_detail(gsplat_renderer* h, const char* id);
void gsplat_renderer_generate_render_geometry(gsplat_renderer* h, GSplatRenderContext* r);
void gsplat_renderer_render(gsplat_renderer* h, GSplatRenderContext* r, int is_object_level);
void gsplat_renderer_post_render(gsplat_renderer* h);
/* ONE redraw as the reference drives it, in one foreign call: includeInRenderPass for each of the `n` ids (GR_PrimGsplat::render,
 * src/GR_GSplat.C:485-492), then generateRenderGeometry -> render -> postRender (MyCustomSceneRenderHook::render,
 * src/DM_GSplatHook.C:30-39).  What bench.py --via-shim times: the cost of the boundary itself, not of one FFI call per verb. */
void gsplat_renderer_redraw(gsplat_renderer* h, const char* const* ids, int n, GSplatRenderContext* r, int is_object_level);
void gsplat_renderer_set_rendering_enabled(gsplat_renderer* h, int enabled);
void gsplat_renderer_set_explicit_camera_pos(gsplat_renderer* h, const float pos[3]);
void gsplat_renderer_set_spherical_harmonics_order(gsplat_renderer* h, int order);
int64_t gsplat_renderer_query(gsplat_renderer* h, int what, const char* id_or_null);
void gsplat_renderer_get_origin(gsplat_renderer* h, float out[3]);
void gsplat_renderer_get_last_camera_pos(gsplat_renderer* h, float out[3]);
gsr_context* gsplat_renderer_engine(gsplat_renderer* h);
gsr_multi* gsplat_renderer_multi(gsplat_renderer* h);
unsigned int gsplat_closest_sqrt_power_of_2(int n);
#ifdef __cplusplus
GSplatRenderer* gsplat_renderer_impl(gsplat_renderer* h);   /* the C++ object behind a handle (C++ callers in the same library) */
#endif

/* ---- attribute ingest (what GR_PrimGsplat::update does before registerUpdate,
 *      src/GR_GSplat.C:302-372): fp32 -> fp16 RNE and the three SH encodings -- */
void gsplat_quantize_half(const float* in, uint16_t* out, int64_t count);
/* sh1..sh15 as 15 arrays of float[3n] (":155-163") -> three half[16n] matrices */
void gsplat_pack_sh_from_vec3(const float* const sh[15], int64_t n, uint16_t* shx, uint16_t* shy, uint16_t* shz);
/* f_rest_0..44, channel-major (":166-177"): 45 arrays of float[n] */
void gsplat_pack_sh_from_frest(const float* const f_rest[45], int64_t n, uint16_t* shx, uint16_t* shy, uint16_t* shz);
/* sh_coefficients float-array attribute, up to 15 vec3 per point, interleaved (":93-113") */
void gsplat_pack_sh_from_array(const float* coeffs, int64_t n, int vec3_per_point, uint16_t* shx, uint16_t* shy, uint16_t* shz);

#ifdef __cplusplus
}
#endif
#endif

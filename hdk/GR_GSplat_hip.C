/*
 * GR_GSplat_hip.C -- GR_PrimGsplat over libgsplat_hip: HDK attribute handles in, GSplatPrim out.
 *
 * Takes the place of the reference's /root/reference/gsplat_plugin/src/GR_GSplat.C.  What moved where:
 *   reference src/GR_GSplat.C                                   here
 *   :93-189   three SH naming schemes (handle discovery)        findShAttributes(): which scheme exists; the mapping
 *                                                               coefficient j -> (row j/4, col j%4) is GSplatPrim's
 *   :233-289  attribute discovery, opacity / Alpha precedence   update(): handles -> float arrays, NULL = absent; the
 *                                                               precedence, the four defaults and the error log are
 *                                                               GSplatPrim::update's (csrc/gsplat_ingest.cpp)
 *   :302-372  tbb loop: gather + fp16 quantisation + SH pack    gather*() below fill float32 arrays in vertex order
 *                                                               (UTparallelFor); GSplatPrim quantises (RNE) and packs
 *   :374-421  wireframe VBOs                                    none: gsr_render_wire draws the overlay from the
 *                                                               resident device arrays (DM_GSplatHook_hip.C)
 *   :423-457  registerUpdate, explicit camera, SH order         GSplatPrim::update
 *   :460-493  render(): enable, include, camera, order          GSplatPrim::render
 *   :63-70    destructor: flushEntriesForMatchingDetail          ~GSplatPrim
 *
 * NOT BUILT IN THIS REPOSITORY (no HDK here: $HFS, hcustom, GA / GR / GT / RE headers); type-checked against stand-ins for the
 * HDK classes it touches (tests/hdk_mock/, tests/test_hdk_glue.py).  Everything below the
 * GSplatPrim calls is exercised by the repo's tests through the same entry points (tests/test_host_shim.py,
 * tests/test_gpu_parity.py::test_ingest_*).  Build: hdk/build.sh.
 */
#include "GR_GSplat_hip.h"

#include <GA/GA_Handle.h>
#include <GT/GT_GEOPrimitive.h>
#include <GR/GR_Utils.h>
#include <GU/GU_Detail.h>
#include <UT/UT_ParallelUtil.h>

#include <cstdio>
#include <vector>

#include "GSplatRenderer.h"     /* this repo: include/ */

GR_Primitive* GR_PrimGsplatHook::createPrimitive(const GT_PrimitiveHandle&, const GEO_Primitive* geo_prim, const GR_RenderInfo* info,
                                                 const char* cache_name, GR_PrimAcceptResult&)
{
    return new GR_PrimGsplat(info, cache_name, geo_prim);
}

GR_PrimGsplat::GR_PrimGsplat(const GR_RenderInfo* info, const char* cache_name, const GEO_Primitive* prim)
    : GR_Primitive(info, cache_name, GA_PrimCompat::TypeMask(0)), myTypeId(prim->getTypeId().get()),
      myPrim(GSplatRenderer::getInstance())
{
}

GR_PrimGsplat::~GR_PrimGsplat() {}

GR_PrimAcceptResult GR_PrimGsplat::acceptPrimitive(GT_PrimitiveType, int geo_type, const GT_PrimitiveHandle&, const GEO_Primitive*)
{
    return geo_type == myTypeId ? GR_PROCESSED : GR_NOT_PROCESSED;
}

namespace {

/* a point attribute of tuple size N as float32 in VERTEX order of the primitive; empty = attribute absent */
template <int N, typename HANDLE>
std::vector<float> gatherTuple(const GEO_PrimGsplat* prim, const GA_Attribute* attr)
{
    std::vector<float> out;
    if (!attr) return out;
    HANDLE h(attr);
    if (!h.isValid()) return out;
    const GA_Size n = prim->getVertexCount();
    out.resize((size_t)n * N);
    UTparallelFor(UT_BlockedRange<GA_Size>(0, n), [&](const UT_BlockedRange<GA_Size>& range) {
        for (GA_Size i = range.begin(); i != range.end(); ++i) {
            const auto v = h.get(prim->getVertexOffset(i));
            for (int k = 0; k < N; ++k) out[(size_t)i * N + k] = (float)v[k];
        }
    });
    return out;
}

std::vector<float> gatherScalar(const GEO_PrimGsplat* prim, const GA_Attribute* attr)
{
    std::vector<float> out;
    if (!attr) return out;
    GA_ROHandleF h(attr);
    if (!h.isValid()) return out;
    const GA_Size n = prim->getVertexCount();
    out.resize((size_t)n);
    UTparallelFor(UT_BlockedRange<GA_Size>(0, n), [&](const UT_BlockedRange<GA_Size>& range) {
        for (GA_Size i = range.begin(); i != range.end(); ++i) out[(size_t)i] = h.get(prim->getVertexOffset(i));
    });
    return out;
}

const float* orNull(const std::vector<float>& v) { return v.empty() ? nullptr : v.data(); }

}  // namespace

void GR_PrimGsplat::update(RE_RenderContext, const GT_PrimitiveHandle& primh, const GR_UpdateParms& p)
{
    const GEO_PrimGsplat* prim = nullptr;
    getGEOPrimFromGT<GEO_PrimGsplat>(primh, prim);
    myHasSplats = prim && prim->getVertexCount() > 0;
    if (!myHasSplats) return;

    GU_DetailHandleAutoReadLock lock(p.geometry);
    const GU_Detail* gdp = lock.getGdp();
    const GA_Size n = prim->getVertexCount();

    /* positions */
    std::vector<float> P((size_t)n * 3);
    UTparallelFor(UT_BlockedRange<GA_Size>(0, n), [&](const UT_BlockedRange<GA_Size>& range) {
        for (GA_Size i = range.begin(); i != range.end(); ++i) {
            const UT_Vector3 v = gdp->getPos3(prim->getVertexOffset(i));
            P[(size_t)i * 3] = v.x(); P[(size_t)i * 3 + 1] = v.y(); P[(size_t)i * 3 + 2] = v.z();
        }
    });
    /* the four per-point attributes: absent -> GSplatPrim applies the reference's default and logs the error once */
    const std::vector<float> Cd = gatherTuple<3, GA_ROHandleV3>(prim, gdp->findPointAttribute("Cd"));
    const std::vector<float> opacity = gatherScalar(prim, gdp->findPointAttribute("opacity"));
    const std::vector<float> Alpha = gatherScalar(prim, gdp->findPointAttribute("Alpha"));     /* wins when present */
    const std::vector<float> scale = gatherTuple<3, GA_ROHandleV3>(prim, gdp->findPointAttribute("scale"));
    const std::vector<float> orient = gatherTuple<4, GA_ROHandleV4>(prim, gdp->findPointAttribute("orient"));

    /* spherical harmonics: the first of the three naming schemes that exists */
    std::vector<float> shArray;                      /* "sh_coefficients": a float array of vec3s per point */
    int shArrayLen = 0;
    std::vector<std::vector<float>> shVec(15), fRest(45);
    const float* shPtr[15] = {nullptr};
    const float* frPtr[45] = {nullptr};
    {
        const GA_Attribute* arr = gdp->findPointAttribute("sh_coefficients");
        GA_ROHandleFA h;
        if (arr && arr->getStorageClass() == GA_STORECLASS_FLOAT && arr->getTupleSize() == 3)
            h = gdp->findFloatArray(GA_ATTRIB_POINT, "sh_coefficients", 0, 15);
        if (h.isValid()) {
            /* the widest array decides the stride; shorter ones are zero-padded (GSplatPrim reads at most 15 vec3) */
            UT_Fpreal32Array vals;
            for (GA_Size i = 0; i < n; ++i) { h.get(prim->getVertexOffset(i), vals); if ((int)vals.size() > shArrayLen) shArrayLen = (int)vals.size(); }
            if (shArrayLen > 45) shArrayLen = 45;
            shArray.assign((size_t)n * shArrayLen, 0.0f);
            for (GA_Size i = 0; i < n; ++i) {
                h.get(prim->getVertexOffset(i), vals);
                for (int k = 0; k < shArrayLen && k < (int)vals.size(); ++k) shArray[(size_t)i * shArrayLen + k] = vals(k);
            }
        } else if (gdp->findPointAttribute("sh1")) {
            char name[16];
            for (int j = 0; j < 15; ++j) {
                std::snprintf(name, sizeof name, "sh%d", j + 1);
                shVec[j] = gatherTuple<3, GA_ROHandleV3>(prim, gdp->findPointAttribute(name));
                if (shVec[j].empty()) break;         /* attributes behind the first gap read as zero */
                shPtr[j] = shVec[j].data();
            }
        } else if (gdp->findPointAttribute("f_rest_0")) {
            char name[16];
            for (int j = 0; j < 45; ++j) {
                std::snprintf(name, sizeof name, "f_rest_%d", j);
                fRest[j] = gatherScalar(prim, gdp->findPointAttribute(name));
                if (fRest[j].empty()) break;
                frPtr[j] = fRest[j].data();
            }
        }
    }

    gsplat_attrs a{};
    a.count = (int64_t)n;
    a.P = P.data();
    a.Cd = orNull(Cd);
    a.opacity = orNull(opacity);
    a.Alpha = orNull(Alpha);
    a.scale = orNull(scale);
    a.orient = orNull(orient);
    a.sh_coefficients = orNull(shArray);
    a.sh_coefficients_len = shArrayLen;
    a.sh = shPtr[0] ? shPtr : nullptr;
    a.f_rest = frPtr[0] ? frPtr : nullptr;

    /* the two detail attributes */
    int32_t order = 3;
    float eye[3] = {0, 0, 0};
    {
        GA_ROHandleI ho(gdp->findAttribute(GA_ATTRIB_GLOBAL, "gsplat__sh_order"));
        if (ho.isValid()) { order = ho.get(GA_Offset(0)); a.sh_order = &order; }
        GA_ROHandleV3 he(gdp->findAttribute(GA_ATTRIB_GLOBAL, "gsplat__explicit_camera_pos"));
        if (he.isValid()) { const UT_Vector3 v = he.get(GA_Offset(0)); eye[0] = v.x(); eye[1] = v.y(); eye[2] = v.z(); a.explicit_camera_pos = eye; }
    }

    /* identity of the primitive: detail pointer, cache version, first vertex offset (the registry id's three parts) */
    GSplatCacheVersion ver;
    for (int k = 0; k < 4; ++k) ver.e[k] = p.geo_version.getElement(k);
    const UT_Vector3 bc = prim->baryCenter();
    const float bary[3] = {bc.x(), bc.y(), bc.z()};
    myPrim.update(gdp, ver, (int64_t)prim->getVertexOffset(0), a, bary);
}

void GR_PrimGsplat::render(RE_RenderContext, GR_RenderMode render_mode, GR_RenderFlags flags, GR_DrawParms)
{
    if (!myHasSplats) return;
    /* wire-over / wireframe display: the scene hook draws the overlay for everything that is marked this redraw */
    extern void GSplatHipRequestWireOverlay(bool);
    GSplatHipRequestWireOverlay(render_mode == GR_RENDER_WIREFRAME || (flags & GR_RENDER_FLAG_WIRE_OVER));
    myPrim.render(render_mode < GR_RENDER_NUM_BEAUTY_MODES);
}

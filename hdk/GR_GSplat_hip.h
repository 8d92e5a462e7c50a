/*
 * GR_GSplat_hip.h -- the viewport primitive of the GSplat plugin over libgsplat_hip (HDK glue).
 *
 * Takes the place of the reference's /root/reference/gsplat_plugin/include/GR_GSplat.h: the same two classes with the
 * same overrides, so GEO_PrimGsplat::registerMyself (src/GEO_GSplat.C:494-498) registers the hook unchanged.  What the
 * reference keeps in eight per-primitive arrays plus a registry id, an explicit camera and an SH order
 * (include/GR_GSplat.h:118-135) lives in ONE GSplatPrim member (include/GSplatPrim.h of this repo).
 *
 * NOT COMPILED IN THIS REPOSITORY: it needs the HDK ($HFS, hcustom).  hdk/build.sh refuses to run without $HFS.
 */
#ifndef GR_GSPLAT_HIP_H
#define GR_GSPLAT_HIP_H

#include <GUI/GUI_PrimitiveHook.h>
#include <GR/GR_Primitive.h>

#include "GEO_GSplat.h"     /* the reference's custom primitive: unchanged */
#include "GSplatPrim.h"     /* this repo: include/ */

class RE_Geometry;

class GR_PrimGsplatHook : public GUI_PrimitiveHook
{
public:
    GR_PrimGsplatHook() : GUI_PrimitiveHook("GSplat") {}
    ~GR_PrimGsplatHook() override {}
    GR_Primitive* createPrimitive(const GT_PrimitiveHandle& gt_prim, const GEO_Primitive* geo_prim, const GR_RenderInfo* info,
                                  const char* cache_name, GR_PrimAcceptResult& processed) override;
};

class GR_PrimGsplat : public GR_Primitive
{
public:
    GR_PrimGsplat(const GR_RenderInfo* info, const char* cache_name, const GEO_Primitive* prim);
    ~GR_PrimGsplat() override;      /* myPrim's destructor flushes the registry entries of this detail */

    const char* className() const override { return "GR_PrimGsplat"; }
    GR_PrimAcceptResult acceptPrimitive(GT_PrimitiveType t, int geo_type, const GT_PrimitiveHandle& ph, const GEO_Primitive* prim) override;
    void update(RE_RenderContext r, const GT_PrimitiveHandle& primh, const GR_UpdateParms& p) override;
    void render(RE_RenderContext r, GR_RenderMode render_mode, GR_RenderFlags flags, GR_DrawParms dp) override;
    void renderDecoration(RE_RenderContext, GR_Decoration, const GR_DecorationParms&) override {}
    int renderPick(RE_RenderContext, const GR_DisplayOption*, unsigned int, GR_PickStyle, bool) override { return 0; }

private:
    int myTypeId;
    bool myHasSplats = false;       /* update() saw a non-empty primitive: render() has something to mark */
    GSplatPrim myPrim;              /* ingest, quantisation, registerUpdate, the per-redraw verbs */
};

#endif

/*
 * hdk_mock.h -- TEST-ONLY stand-ins for the ~40 HDK classes / enums that hdk/GR_GSplat_hip.C and hdk/DM_GSplatHook_hip.C touch,
 * declared just far enough for a C++ front end to TYPE-CHECK the glue (tests/test_hdk_glue.py: hipcc -fsyntax-only).
 *
 * Written from the way the glue USES the HDK (and the way the reference's src/GR_GSplat.C:191-493 and src/DM_GSplatHook.C:30-73 do):
 * it pins nothing about Houdini's real headers -- signatures here are the glue's own assumptions -- but it catches what a regular
 * expression cannot: misspelt members, wrong argument counts, const errors, missing includes, overrides that override nothing.
 * Never shipped, never linked; the reference's sources are not pulled in (GEO_GSplat.h below is a stub of the one class used).
 */
#ifndef GSPLAT_TEST_HDK_MOCK_H
#define GSPLAT_TEST_HDK_MOCK_H
#include <cstddef>
#include <cstdint>

/* ---- UT ----------------------------------------------------------------- */
typedef float fpreal32;
class UT_Vector3 {
public:
    float x() const; float y() const; float z() const;
    float operator[](int) const;
};
class UT_Vector4 {
public:
    float operator[](int) const;
};
class UT_Matrix4D {};
class UT_Matrix4F {
public:
    UT_Matrix4F();
    explicit UT_Matrix4F(float diagonal);
    UT_Matrix4F(const UT_Matrix4D&);
    const float* data() const;
};
class UT_Fpreal32Array {
public:
    int64_t size() const;
    float operator()(int64_t i) const;
};
template <typename T> class UT_BlockedRange {
public:
    UT_BlockedRange(T b, T e);
    T begin() const; T end() const;
};
template <typename RANGE, typename BODY> void UTparallelFor(const RANGE& r, const BODY& body) { body(r); }

/* ---- GA / GEO / GU ------------------------------------------------------- */
typedef int64_t GA_Size;
class GA_Offset {
public:
    GA_Offset() {}
    explicit GA_Offset(int64_t) {}
    operator int64_t() const;
};
enum GA_AttributeOwner { GA_ATTRIB_VERTEX, GA_ATTRIB_POINT, GA_ATTRIB_PRIMITIVE, GA_ATTRIB_GLOBAL };
enum GA_StorageClass { GA_STORECLASS_INT, GA_STORECLASS_FLOAT };
namespace GA_PrimCompat { struct TypeMask { explicit TypeMask(int) {} }; }
class GA_Attribute {
public:
    GA_StorageClass getStorageClass() const;
    int getTupleSize() const;
};
template <typename T> class GA_ROHandleT {
public:
    GA_ROHandleT();
    GA_ROHandleT(const GA_Attribute*);
    bool isValid() const;
    T get(GA_Offset) const;
};
typedef GA_ROHandleT<float> GA_ROHandleF;
typedef GA_ROHandleT<int> GA_ROHandleI;
typedef GA_ROHandleT<UT_Vector3> GA_ROHandleV3;
typedef GA_ROHandleT<UT_Vector4> GA_ROHandleV4;
class GA_ROHandleFA {
public:
    GA_ROHandleFA();
    bool isValid() const;
    void get(GA_Offset, UT_Fpreal32Array&) const;
};
class GA_PrimitiveTypeId { public: int get() const; };
class GEO_Primitive {
public:
    virtual ~GEO_Primitive();
    const GA_PrimitiveTypeId& getTypeId() const;
    GA_Size getVertexCount() const;
    GA_Offset getVertexOffset(GA_Size i) const;
    UT_Vector3 baryCenter() const;
};
class GU_Detail {
public:
    UT_Vector3 getPos3(GA_Offset) const;
    const GA_Attribute* findPointAttribute(const char* name) const;
    const GA_Attribute* findAttribute(GA_AttributeOwner, const char* name) const;
    GA_ROHandleFA findFloatArray(GA_AttributeOwner, const char* name, int min_size, int max_size) const;
};
class GU_ConstDetailHandle {};
class GU_DetailHandleAutoReadLock {
public:
    explicit GU_DetailHandleAutoReadLock(const GU_ConstDetailHandle&);
    const GU_Detail* getGdp() const;
};

/* ---- GT ----------------------------------------------------------------- */
typedef int GT_PrimitiveType;
class GT_Primitive;
class GT_PrimitiveHandle {};
template <typename T> void getGEOPrimFromGT(const GT_PrimitiveHandle&, const T*& out);

/* ---- RE ----------------------------------------------------------------- */
enum RE_BlendSourceFactor { RE_SBLEND_ONE, RE_SBLEND_ONE_MINUS_DST_ALPHA };
enum RE_BlendDestFactor { RE_DBLEND_ZERO, RE_DBLEND_ONE };
class RE_Render {
public:
    void getMatrix(UT_Matrix4D&) const;
    void getProjectionMatrix(UT_Matrix4D&) const;
    void pushBlendState(); void popBlendState();
    void pushDepthState(); void popDepthState();
    void blend(int on);
    void setBlendFunction(RE_BlendSourceFactor, RE_BlendDestFactor);
    void setAlphaBlendFunction(RE_BlendSourceFactor, RE_BlendDestFactor);
    void disableDepthTest();
    void disableDepthBufferWriting();
};
class RE_RenderContext {
public:
    RE_Render* operator->() const;
};
class RE_CacheVersion { public: int64_t getElement(int k) const; };

/* ---- GR / GUI ------------------------------------------------------------ */
class GR_RenderInfo;
class GR_DisplayOption;
class GR_DecorationParms;
enum GR_PrimAcceptResult { GR_NOT_PROCESSED, GR_PROCESSED, GR_PROCESSED_NON_EXCLUSIVE };
enum GR_RenderMode { GR_RENDER_BEAUTY, GR_RENDER_MATERIAL, GR_RENDER_NUM_BEAUTY_MODES, GR_RENDER_WIREFRAME, GR_RENDER_HIDDEN_LINE };
enum GR_RenderFlags { GR_RENDER_FLAG_NONE = 0, GR_RENDER_FLAG_WIRE_OVER = 4 };
enum GR_Decoration { GR_NO_DECORATION };
enum GR_PickStyle { GR_PICK_NONE };
struct GR_DrawParms {};
struct GR_UpdateParms {
    GU_ConstDetailHandle geometry;
    RE_CacheVersion geo_version;
};
class GR_Primitive {
public:
    GR_Primitive(const GR_RenderInfo*, const char* cache_name, GA_PrimCompat::TypeMask);
    virtual ~GR_Primitive();
    virtual const char* className() const = 0;
    virtual GR_PrimAcceptResult acceptPrimitive(GT_PrimitiveType, int geo_type, const GT_PrimitiveHandle&, const GEO_Primitive*) = 0;
    virtual void update(RE_RenderContext, const GT_PrimitiveHandle&, const GR_UpdateParms&) = 0;
    virtual void render(RE_RenderContext, GR_RenderMode, GR_RenderFlags, GR_DrawParms) = 0;
    virtual void renderDecoration(RE_RenderContext, GR_Decoration, const GR_DecorationParms&);
    virtual int renderPick(RE_RenderContext, const GR_DisplayOption*, unsigned int, GR_PickStyle, bool) = 0;
};
class GUI_PrimitiveHook {
public:
    explicit GUI_PrimitiveHook(const char* name);
    virtual ~GUI_PrimitiveHook();
    virtual GR_Primitive* createPrimitive(const GT_PrimitiveHandle&, const GEO_Primitive*, const GR_RenderInfo*, const char* cache_name,
                                          GR_PrimAcceptResult& processed);
};
class GUI_DisplayOption { public: bool isObjectLevel() const; };

/* ---- DM ----------------------------------------------------------------- */
class DM_VPortAgent;
enum DM_ViewportType { DM_VIEWPORT_PERSPECTIVE = 1, DM_VIEWPORT_ALL = 0xff };
enum DM_SceneHookType { DM_HOOK_BACKGROUND, DM_HOOK_BEAUTY, DM_HOOK_FOREGROUND };
enum DM_SceneHookPolicy { DM_HOOK_BEFORE_NATIVE, DM_HOOK_AFTER_NATIVE, DM_HOOK_REPLACE_NATIVE };
enum DM_SceneHookViews { DM_HOOK_ALL_VIEWS };
struct DM_SceneHookData {
    int view_width, view_height;
    const GUI_DisplayOption* disp_options;
};
class DM_SceneRenderHook {
public:
    DM_SceneRenderHook(DM_VPortAgent&, DM_ViewportType view_mask);
    virtual ~DM_SceneRenderHook();
    virtual bool render(RE_RenderContext r, const DM_SceneHookData& hook_data) = 0;
};
class DM_SceneHook {
public:
    DM_SceneHook(const char* name, int priority, DM_SceneHookViews);
    virtual ~DM_SceneHook();
    virtual DM_SceneRenderHook* newSceneRender(DM_VPortAgent&, DM_SceneHookType, DM_SceneHookPolicy) = 0;
    virtual void retireSceneRender(DM_VPortAgent&, DM_SceneRenderHook*) = 0;
};
class DM_RenderTable {
public:
    bool registerSceneHook(DM_SceneHook*, DM_SceneHookType, DM_SceneHookPolicy);
};

#endif

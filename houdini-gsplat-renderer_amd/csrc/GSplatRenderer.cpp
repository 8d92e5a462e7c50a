// GSplatRenderer.cpp -- HDK-free host shim: registry / active-set diff / origin /
// concatenation with the 2^23-1 cap / camera position / SH-order gate, then the
// libgsplat_hip engine.  Mirrors, verb by verb,
// /root/reference/gsplat_plugin/src/GSplatRenderer.C:218-320 (registry),
// :322-532 (generateRenderGeometry), :534-658 (render), :660-694 (postRender and
// setters).  What is NOT mirrored: GL textures, shader manager, RE_Geometry.
#include "../../include/GSplatRenderer.h"

#include <algorithm>
#include <cinttypes>
#include <cstdarg>
#include <new>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <sstream>

namespace {

void logLine(const char* level, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    std::fprintf(stderr, "[GSplat %s] ", level);
    std::vfprintf(stderr, fmt, ap);
    std::fputc('\n', stderr);
    va_end(ap);
}

// general 4x4 inverse in double (UT_Matrix4D::invert stand-in); m and out are
// 16 doubles in the same memory order.  Returns false if singular.
bool invert4(const double* m, double* out)
{
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            a[r][c] = m[r * 4 + c];
            a[r][4 + c] = (r == c) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0) return false;
        if (piv != col)
            for (int c = 0; c < 8; ++c) std::swap(a[piv][c], a[col][c]);
        const double d = a[col][col];
        for (int c = 0; c < 8; ++c) a[col][c] /= d;
        for (int r = 0; r < 4; ++r)
            if (r != col) {
                const double fct = a[r][col];
                if (fct != 0.0)
                    for (int c = 0; c < 8; ++c) a[r][c] -= fct * a[col][c];
            }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[r * 4 + c] = a[r][4 + c];
    return true;
}

}  // namespace

GSplatRenderer& GSplatRenderer::getInstance()
{
    static GSplatRenderer instance(0);
    return instance;
}

GSplatRenderer::GSplatRenderer(int device)
{
    if (device < 0) {
        myDry = true;
        return;
    }
    myLastStatus = gsr_create(device, &myEngine);
    if (myLastStatus != GSR_OK) {
        logLine("ERROR", "GPU engine unavailable: %s", gsr_last_error());
        myEngine = nullptr;
    }
}

GSplatRenderer::~GSplatRenderer()
{
    if (myEngine) gsr_destroy(myEngine);
}

unsigned int GSplatRenderer::closestSqrtPowerOf2(int n)
{
    if (n <= 1) return 2;
    const float sqrtVal = std::sqrt(static_cast<float>(n));
    const unsigned int power = static_cast<unsigned int>(std::ceil(std::log2(sqrtVal)));
    return 1u << power;
}

// src/GSplatRenderer.C:218-291
std::string GSplatRenderer::registerUpdate(const void* gdp, const GSplatCacheVersion& gversion, int64_t gvtx,
                                           int64_t splatCount, const float splatOrigin[3], const float* splatPts,
                                           const uint16_t* splatColors, const float* splatAlphas,
                                           const uint16_t* splatScales, const uint16_t* splatOrients,
                                           const uint16_t* splatShxs, const uint16_t* splatShys,
                                           const uint16_t* splatShzs, int64_t shCount)
{
    if (!myVersionLogged) {
        logLine("INFO", "%s", gsr_version());
        myVersionLogged = true;
    }
    std::ostringstream oss;
    oss << std::hex << std::showbase << reinterpret_cast<uintptr_t>(gdp) << "__" << std::dec << gvtx << "__"
        << gversion.e[0] << "_" << gversion.e[1] << "_" << gversion.e[2] << "_" << gversion.e[3];
    const std::string registryId = oss.str();

    // entries of the same detail with another cache version are stale (:246-265)
    for (auto it = myRenderStateRegistry.begin(); it != myRenderStateRegistry.end();) {
        if (it->second->gdp == gdp && it->second->gversion != gversion)
            it = myRenderStateRegistry.erase(it);
        else
            ++it;
    }
    auto& slot = myRenderStateRegistry[registryId];
    if (!slot) slot.reset(new GSplatRegisterEntry());
    GSplatRegisterEntry& e = *slot;
    e.gversion = gversion;
    e.gdp = gdp;
    e.gvtx = gvtx;
    for (int k = 0; k < 3; ++k) e.splatOrigin[k] = splatOrigin ? splatOrigin[k] : 0.0f;
    e.splatPts = splatPts;
    e.splatColors = splatColors;
    e.splatAlphas = splatAlphas;
    e.splatScales = splatScales;
    e.splatOrients = splatOrients;
    e.splatShxs = splatShxs;
    e.splatShys = splatShys;
    e.splatShzs = splatShzs;
    e.shCount = (splatShxs && splatShys && splatShzs) ? shCount : 0;
    e.splatCount = splatCount;
    e.active = false;
    e.age = -1;
    e.ageSinceLastActive = -1;
    return registryId;
}

// :293-311
void GSplatRenderer::flushEntriesForMatchingDetail(const std::string& registryId)
{
    auto it = myRenderStateRegistry.find(registryId);
    if (it == myRenderStateRegistry.end()) return;
    const void* gdp = it->second->gdp;
    for (auto jt = myRenderStateRegistry.begin(); jt != myRenderStateRegistry.end();) {
        if (jt->second->gdp == gdp)
            jt = myRenderStateRegistry.erase(jt);
        else
            ++jt;
    }
}

// :313-320
void GSplatRenderer::includeInRenderPass(const std::string& registryId)
{
    auto it = myRenderStateRegistry.find(registryId);
    if (it != myRenderStateRegistry.end()) it->second->active = true;
}

// :141-153
bool GSplatRenderer::isRenderStateRegistryCurrent() const
{
    std::set<std::string> requested;
    for (const auto& kv : myRenderStateRegistry)
        if (kv.second->active) requested.insert(kv.first);
    return myActiveRegistries == requested;
}

// :322-532.  The TBB pack into GL textures becomes gsr_upload_begin/append/end
// (device-side repack into SoA); everything else keeps the reference's logic.
void GSplatRenderer::generateRenderGeometry(GSplatRenderContext& /*r*/)
{
    if (isRenderStateRegistryCurrent()) return;

    const int64_t GSplatCountMax = GSPLAT_COUNT_MAX - 1;
    myActiveRegistries.clear();
    int64_t totalSplatCount = 0;
    bool isShDataPresent = true;
    myCanRender = false;
    bool isGsplatCapHit = false;
    int64_t totalActiveSplats = 0;
    for (const auto& kv : myRenderStateRegistry) {
        totalActiveSplats += kv.second->splatCount;
        if (!isGsplatCapHit && kv.second->active && kv.second->splatCount > 0) {
            myActiveRegistries.insert(kv.first);
            totalSplatCount += kv.second->splatCount;
            isShDataPresent = kv.second->shCount > 0;  // value of the LAST active entry (:353)
        }
        if (totalSplatCount >= GSplatCountMax) isGsplatCapHit = true;
    }
    if (!totalSplatCount) return;

    myGSplatCount = std::min(totalSplatCount, GSplatCountMax);
    if (isGsplatCapHit)
        logLine("WARNING", "%" PRId64 " active GSplats, exceeds %" PRId64 " budget. Culling excess %" PRId64 " GSplats!",
                totalActiveSplats, GSplatCountMax, totalActiveSplats - myGSplatCount);
    myCanRender = true;
    myIsShDataPresent = isShDataPresent;

    // origin = mean of the active entries' barycentres (:403-418)
    mySplatOrigin[0] = mySplatOrigin[1] = mySplatOrigin[2] = 0.0f;
    int splatClusters = 0;
    for (const auto& id : myActiveRegistries) {
        const GSplatRegisterEntry* entry = myRenderStateRegistry[id].get();
        if (entry) {
            for (int k = 0; k < 3; ++k) mySplatOrigin[k] += entry->splatOrigin[k];
            ++splatClusters;
        }
    }
    if (splatClusters > 0)
        for (int k = 0; k < 3; ++k) mySplatOrigin[k] /= static_cast<float>(splatClusters);

    ++myStagingCount;
    if (myDry) return;
    if (!myEngine) { myCanRender = false; return; }

    myLastStatus = gsr_upload_begin(myEngine, myGSplatCount, myIsShDataPresent ? 1 : 0, mySplatOrigin);
    if (myLastStatus != GSR_OK) {
        logLine("ERROR", "staging failed: %s", gsr_last_error());
        myCanRender = false;
        return;
    }
    int64_t offset = 0;
    std::vector<uint16_t> zeros;  // for entries without SH when the pass carries SH (the reference would
                                  // index empty arrays there; zero coefficients are the safe reading)
    for (const auto& id : myActiveRegistries) {
        const GSplatRegisterEntry* entry = myRenderStateRegistry[id].get();
        if (!entry) continue;
        int64_t splatCount = entry->splatCount;
        const int64_t budgetLeft = GSplatCountMax - offset;
        if (budgetLeft <= 0) break;
        splatCount = std::min(splatCount, budgetLeft);
        const uint16_t *sx = entry->splatShxs, *sy = entry->splatShys, *sz = entry->splatShzs;
        if (myIsShDataPresent && entry->shCount < splatCount) {
            zeros.assign(static_cast<size_t>(splatCount) * 16, 0);
            sx = sy = sz = zeros.data();
        }
        myLastStatus = gsr_upload_append(myEngine, splatCount, entry->splatPts, entry->splatColors, entry->splatAlphas,
                                         entry->splatScales, entry->splatOrients, myIsShDataPresent ? sx : nullptr,
                                         myIsShDataPresent ? sy : nullptr, myIsShDataPresent ? sz : nullptr);
        if (myLastStatus != GSR_OK) break;
        offset += splatCount;
        if (offset >= GSplatCountMax) break;
    }
    if (myLastStatus == GSR_OK) myLastStatus = gsr_upload_end(myEngine);
    if (myLastStatus != GSR_OK) {
        logLine("ERROR", "staging failed: %s", gsr_last_error());
        myCanRender = false;
    }
}

// :534-658
void GSplatRenderer::render(GSplatRenderContext& r, bool isObjectLevel)
{
    if (!myIsRenderEnabled || !myCanRender) return;
    bool anythingRenderable = false;
    for (const auto& kv : myRenderStateRegistry) anythingRenderable |= kv.second->active;
    if (!anythingRenderable) return;

    float camera_pos[3];
    if (myIsExplicitCameraPosSet) {
        for (int k = 0; k < 3; ++k) camera_pos[k] = myExplicitCameraPos[k];
    } else {
        // rowVecMult(0, inverse(view)) = translation row of the inverse (:556-562)
        double vm[16], inv[16];
        for (int k = 0; k < 16; ++k) vm[k] = r.view[k];
        if (!invert4(vm, inv)) return;
        for (int k = 0; k < 3; ++k) camera_pos[k] = static_cast<float>(inv[12 + k]);
    }
    for (int k = 0; k < 3; ++k) myLastCameraPos[k] = camera_pos[k];

    if (isObjectLevel) {
        if (!myJustPrintedOBJLevelRenderingWarning) {
            logLine("WARNING",
                    "Rendering OBJ context with camera position (%3f, %3f, %3f). Note that OBJ transforms different "
                    "to identity are not currently supported (results might appear incorrect).",
                    camera_pos[0], camera_pos[1], camera_pos[2]);
            myJustPrintedOBJLevelRenderingWarning = true;
        }
    } else {
        myJustPrintedOBJLevelRenderingWarning = false;
    }

    if (myDry) { ++myRenderCount; return; }
    if (!myEngine || !r.target) return;

    const bool doSH = (myShOrder > 0 && myIsShDataPresent);  // :623
    gsr_camera cam;
    std::memcpy(cam.obj_view, r.obj_view, sizeof(cam.obj_view));
    std::memcpy(cam.object, r.object, sizeof(cam.object));
    std::memcpy(cam.inv_object, r.inv_object, sizeof(cam.inv_object));
    std::memcpy(cam.view, r.view, sizeof(cam.view));
    std::memcpy(cam.proj, r.proj, sizeof(cam.proj));
    for (int k = 0; k < 3; ++k) cam.cam_pos[k] = camera_pos[k];
    cam.width = r.width;
    cam.height = r.height;
    cam.sh_order = doSH ? myShOrder : 0;
    // the engine re-sorts only when cam_pos or the geometry changed: argsortByDistance's
    // caching with threshold 0 (:165-186)
    myLastStatus = gsr_render_depth(myEngine, &cam, r.depth, r.depth_is_device, r.target, r.target_is_device);
    if (myLastStatus != GSR_OK) {
        logLine("ERROR", "render failed: %s", gsr_last_error());
        return;
    }
    ++myRenderCount;
}

// :660-678
void GSplatRenderer::postRender()
{
    for (auto& kv : myRenderStateRegistry) {
        GSplatRegisterEntry& e = *kv.second;
        if (e.active)
            e.ageSinceLastActive = 0;
        else if (e.ageSinceLastActive > -1)
            ++e.ageSinceLastActive;
        e.active = false;
        ++e.age;
    }
    myIsExplicitCameraPosSet = false;
}

void GSplatRenderer::setRenderingEnabled(bool isRenderEnabled) { myIsRenderEnabled = isRenderEnabled; }

void GSplatRenderer::setExplicitCameraPos(const float p[3])
{
    myIsExplicitCameraPosSet = true;
    for (int k = 0; k < 3; ++k) myExplicitCameraPos[k] = p[k];
}

void GSplatRenderer::setSphericalHarmonicsOrder(int shOrder) { myShOrder = shOrder; }

int64_t GSplatRenderer::query(int what, const std::string& id) const
{
    switch (what) {
    case Q_REGISTRY_SIZE: return static_cast<int64_t>(myRenderStateRegistry.size());
    case Q_ACTIVE_STAGED: return static_cast<int64_t>(myActiveRegistries.size());
    case Q_SPLAT_COUNT: return myGSplatCount;
    case Q_CAN_RENDER: return myCanRender ? 1 : 0;
    case Q_STAGING_COUNT: return myStagingCount;
    case Q_RENDER_COUNT: return myRenderCount;
    case Q_SH_PRESENT: return myIsShDataPresent ? 1 : 0;
    case Q_LAST_STATUS: return myLastStatus;
    case Q_ENTRY_AGE:
    case Q_ENTRY_AGE_SINCE_ACTIVE: {
        auto it = myRenderStateRegistry.find(id);
        if (it == myRenderStateRegistry.end()) return -1000;
        return what == Q_ENTRY_AGE ? it->second->age : it->second->ageSinceLastActive;
    }
    default: return -1;
    }
}

void GSplatRenderer::origin(float out[3]) const { for (int k = 0; k < 3; ++k) out[k] = mySplatOrigin[k]; }
void GSplatRenderer::lastCameraPos(float out[3]) const { for (int k = 0; k < 3; ++k) out[k] = myLastCameraPos[k]; }

// ---------------------------------------------------------------------------
// flat C wrappers
struct gsplat_renderer {
    GSplatRenderer* impl;
    bool owned;
};

extern "C" {

gsplat_renderer* gsplat_renderer_create(int device)
{
    GSplatRenderer* p = new (std::nothrow) GSplatRenderer(device);
    if (!p) return nullptr;
    if (device >= 0 && !p->engine()) { delete p; return nullptr; }
    gsplat_renderer* h = new (std::nothrow) gsplat_renderer{p, true};
    if (!h) delete p;
    return h;
}

gsplat_renderer* gsplat_renderer_get_instance(void)
{
    static gsplat_renderer single{&GSplatRenderer::getInstance(), false};
    return &single;
}

void gsplat_renderer_destroy(gsplat_renderer* h)
{
    if (!h || !h->owned) return;
    delete h->impl;
    delete h;
}

int gsplat_renderer_register_update(gsplat_renderer* h, uint64_t gdp, const int64_t gversion[4], int64_t gvtx_offset,
                                    int64_t splat_count, const float origin[3], const float* P, const uint16_t* Cd,
                                    const float* alpha, const uint16_t* scale, const uint16_t* orient,
                                    const uint16_t* shx, const uint16_t* shy, const uint16_t* shz, int64_t sh_count,
                                    char* id_out, int id_cap)
{
    if (!h || !gversion) return -1;
    GSplatCacheVersion v;
    for (int k = 0; k < 4; ++k) v.e[k] = gversion[k];
    const std::string id = h->impl->registerUpdate(reinterpret_cast<const void*>(static_cast<uintptr_t>(gdp)), v,
                                                   gvtx_offset, splat_count, origin, P, Cd, alpha, scale, orient, shx,
                                                   shy, shz, sh_count);
    if (id_out && id_cap > 0) {
        std::strncpy(id_out, id.c_str(), static_cast<size_t>(id_cap) - 1);
        id_out[id_cap - 1] = '\0';
    }
    return static_cast<int>(id.size());
}

void gsplat_renderer_include_in_render_pass(gsplat_renderer* h, const char* id) { if (h && id) h->impl->includeInRenderPass(id); }
void gsplat_renderer_flush_entries_for_matching_detail(gsplat_renderer* h, const char* id) { if (h && id) h->impl->flushEntriesForMatchingDetail(id); }
void gsplat_renderer_generate_render_geometry(gsplat_renderer* h, GSplatRenderContext* r) { if (h && r) h->impl->generateRenderGeometry(*r); }
void gsplat_renderer_render(gsplat_renderer* h, GSplatRenderContext* r, int is_object_level) { if (h && r) h->impl->render(*r, is_object_level != 0); }
void gsplat_renderer_post_render(gsplat_renderer* h) { if (h) h->impl->postRender(); }
void gsplat_renderer_set_rendering_enabled(gsplat_renderer* h, int enabled) { if (h) h->impl->setRenderingEnabled(enabled != 0); }
void gsplat_renderer_set_explicit_camera_pos(gsplat_renderer* h, const float pos[3]) { if (h && pos) h->impl->setExplicitCameraPos(pos); }
void gsplat_renderer_set_spherical_harmonics_order(gsplat_renderer* h, int order) { if (h) h->impl->setSphericalHarmonicsOrder(order); }
int64_t gsplat_renderer_query(gsplat_renderer* h, int what, const char* id) { return h ? h->impl->query(what, id ? std::string(id) : std::string()) : -1; }
void gsplat_renderer_get_origin(gsplat_renderer* h, float out[3]) { if (h && out) h->impl->origin(out); }
void gsplat_renderer_get_last_camera_pos(gsplat_renderer* h, float out[3]) { if (h && out) h->impl->lastCameraPos(out); }
gsr_context* gsplat_renderer_engine(gsplat_renderer* h) { return h ? h->impl->engine() : nullptr; }
unsigned int gsplat_closest_sqrt_power_of_2(int n) { return GSplatRenderer::closestSqrtPowerOf2(n); }

// ---- ingest: fp32 -> fp16 (round to nearest even, overflow to inf), SH packing
static inline uint16_t f2h(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return static_cast<uint16_t>(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u : 0u));
    if (ax >= 0x477ff000u) return static_cast<uint16_t>(sign | 0x7c00u);
    if (ax < 0x33000001u) return static_cast<uint16_t>(sign);
    const int e = static_cast<int>(ax >> 23) - 127;
    const uint32_t man = (ax & 0x7fffffu) | 0x800000u;
    const int shift = (e < -14) ? 13 + (-14 - e) : 13;
    uint32_t kept = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (kept & 1u))) ++kept;
    const uint32_t out = (e < -14) ? kept : ((static_cast<uint32_t>(e + 15) - 1u) << 10) + kept;
    return static_cast<uint16_t>(sign | out);
}

void gsplat_quantize_half(const float* in, uint16_t* out, int64_t count)
{
    for (int64_t i = 0; i < count; ++i) out[i] = f2h(in[i]);
}

// coefficient j (0-based, = sh(j+1)) goes to (row j/4, col j%4) of a zero-initialised
// row-major 4x4 (src/GR_GSplat.C:322-353)
void gsplat_pack_sh_from_vec3(const float* const sh[15], int64_t n, uint16_t* shx, uint16_t* shy, uint16_t* shz)
{
    for (int64_t i = 0; i < n; ++i) {
        for (int j = 0; j < 16; ++j) shx[16 * i + j] = shy[16 * i + j] = shz[16 * i + j] = 0;
        for (int j = 0; j < 15; ++j) {
            shx[16 * i + j] = f2h(sh[j][3 * i + 0]);
            shy[16 * i + j] = f2h(sh[j][3 * i + 1]);
            shz[16 * i + j] = f2h(sh[j][3 * i + 2]);
        }
    }
}

void gsplat_pack_sh_from_frest(const float* const f_rest[45], int64_t n, uint16_t* shx, uint16_t* shy, uint16_t* shz)
{
    for (int64_t i = 0; i < n; ++i) {
        shx[16 * i + 15] = shy[16 * i + 15] = shz[16 * i + 15] = 0;
        for (int j = 0; j < 15; ++j) {  // (f_rest_j, f_rest_{j+15}, f_rest_{j+30})  (:357-367)
            shx[16 * i + j] = f2h(f_rest[j][i]);
            shy[16 * i + j] = f2h(f_rest[j + 15][i]);
            shz[16 * i + j] = f2h(f_rest[j + 30][i]);
        }
    }
}

void gsplat_pack_sh_from_array(const float* coeffs, int64_t n, int vec3_per_point, uint16_t* shx, uint16_t* shy, uint16_t* shz)
{
    const int m = vec3_per_point < 0 ? 0 : (vec3_per_point > 16 ? 16 : vec3_per_point);
    for (int64_t i = 0; i < n; ++i) {
        for (int j = 0; j < 16; ++j) shx[16 * i + j] = shy[16 * i + j] = shz[16 * i + j] = 0;
        for (int j = 0; j < m; ++j) {  // (:330-340)
            const float* v = coeffs + (static_cast<size_t>(i) * vec3_per_point + j) * 3;
            shx[16 * i + j] = f2h(v[0]);
            shy[16 * i + j] = f2h(v[1]);
            shz[16 * i + j] = f2h(v[2]);
        }
    }
}

}  // extern "C"

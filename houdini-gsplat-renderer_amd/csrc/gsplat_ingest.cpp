// gsplat_ingest.cpp -- what happens to point attributes BEFORE GSplatRenderer::registerUpdate:
// fp32 -> fp16 quantisation, the three SH naming schemes, attribute precedence and defaults, and the GSplatPrim
// class that plays GR_PrimGsplat's part of the frame protocol (include/GSplatPrim.h).
// Behaviour follows /root/reference/gsplat_plugin/src/GR_GSplat.C:93-189 (SH discovery), :233-289 (attribute lookup,
// Alpha over opacity), :302-372 (defaults, quantisation, SH slot mapping), :438-457 (detail attributes), :472-492
// (per-redraw verbs) and src/GEO_GSplat.C:338-351 (barycentre).
#include "../../include/GSplatPrim.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

namespace {

// The reference quantises and packs in a tbb::parallel_for over the points (src/GR_GSplat.C:302-372).  Same here with plain
// threads: fn(begin, end) over disjoint ranges of [0, n); small inputs stay on the calling thread.
template <typename F>
void parallel_ranges(int64_t n, int64_t grain, F&& fn)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int64_t parts = std::min<int64_t>(std::min<int64_t>(hw, 64), (n + grain - 1) / grain);
    if (parts <= 1) { fn(static_cast<int64_t>(0), n); return; }
    std::vector<std::thread> pool;
    pool.reserve(static_cast<size_t>(parts));
    const int64_t step = (n + parts - 1) / parts;
    for (int64_t b = 0; b < n; b += step) pool.emplace_back([&fn, b, step, n] { fn(b, std::min(n, b + step)); });
    for (auto& t : pool) t.join();
}

// binary32 -> binary16, round to nearest even, overflow to infinity (what HDK's fpreal16 constructors do)
inline uint16_t to_half(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t mag = x & 0x7fffffffu;
    if (mag >= 0x7f800000u) return static_cast<uint16_t>(sign | 0x7c00u | (mag > 0x7f800000u ? 0x200u : 0u));   // inf, NaN
    if (mag >= 0x477ff000u) return static_cast<uint16_t>(sign | 0x7c00u);                                       // >= 65520
    if (mag < 0x33000001u) return static_cast<uint16_t>(sign);                                                  // <= 2^-25
    const int exp = static_cast<int>(mag >> 23) - 127;
    const uint32_t sig = (mag & 0x7fffffu) | 0x800000u;
    const int drop = exp < -14 ? 13 + (-14 - exp) : 13;          // mantissa bits that do not fit
    uint32_t q = sig >> drop;
    const uint32_t rest = sig & ((1u << drop) - 1u), half = 1u << (drop - 1);
    if (rest > half || (rest == half && (q & 1u))) ++q;
    const uint32_t bits = exp < -14 ? q : ((static_cast<uint32_t>(exp + 15) - 1u) << 10) + q;   // (q carries the hidden bit)
    return static_cast<uint16_t>(sign | bits);
}

void complain(const char* what)
{
    std::fprintf(stderr, "gsplat-hip error: %s\n", what);
}

}  // namespace

extern "C" {

void gsplat_quantize_half(const float* in, uint16_t* out, int64_t count)
{
    parallel_ranges(count, 1 << 16, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) out[i] = to_half(in[i]); });
}

// The reference keeps SH in three 4x4 half matrices per splat (x/y/z channel), coefficient j (= sh(j+1)) in flat
// slot j; slot 15 stays 0.
void gsplat_pack_sh_from_vec3(const float* const sh[15], int64_t n, uint16_t* shx, uint16_t* shy, uint16_t* shz)
{
    parallel_ranges(n, 1 << 13, [&](int64_t b_, int64_t e_) {
    for (int64_t i = b_; i < e_; ++i) {
        uint16_t *x = shx + 16 * i, *y = shy + 16 * i, *z = shz + 16 * i;
        for (int j = 0; j < 16; ++j) {
            const float* v = (j < 15 && sh[j]) ? sh[j] + 3 * i : nullptr;
            x[j] = v ? to_half(v[0]) : 0;
            y[j] = v ? to_half(v[1]) : 0;
            z[j] = v ? to_half(v[2]) : 0;
        }
    }
    });
}

void gsplat_pack_sh_from_frest(const float* const f_rest[45], int64_t n, uint16_t* shx, uint16_t* shy, uint16_t* shz)
{
    parallel_ranges(n, 1 << 13, [&](int64_t b_, int64_t e_) {
    for (int64_t i = b_; i < e_; ++i) {
        uint16_t *x = shx + 16 * i, *y = shy + 16 * i, *z = shz + 16 * i;
        for (int j = 0; j < 15; ++j) {   // channel-major INRIA layout: (f_rest_j, f_rest_{j+15}, f_rest_{j+30})
            x[j] = f_rest[j] ? to_half(f_rest[j][i]) : 0;
            y[j] = f_rest[j + 15] ? to_half(f_rest[j + 15][i]) : 0;
            z[j] = f_rest[j + 30] ? to_half(f_rest[j + 30][i]) : 0;
        }
        x[15] = y[15] = z[15] = 0;
    }
    });
}

void gsplat_pack_sh_from_array(const float* coeffs, int64_t n, int vec3_per_point, uint16_t* shx, uint16_t* shy, uint16_t* shz)
{
    const int have = vec3_per_point < 0 ? 0 : vec3_per_point;
    const int used = have > 16 ? 16 : have;
    parallel_ranges(n, 1 << 13, [&](int64_t b_, int64_t e_) {
    for (int64_t i = b_; i < e_; ++i) {
        const float* v = coeffs + static_cast<size_t>(i) * have * 3;
        uint16_t *x = shx + 16 * i, *y = shy + 16 * i, *z = shz + 16 * i;
        for (int j = 0; j < 16; ++j) {
            x[j] = j < used ? to_half(v[3 * j]) : 0;
            y[j] = j < used ? to_half(v[3 * j + 1]) : 0;
            z[j] = j < used ? to_half(v[3 * j + 2]) : 0;
        }
    }
    });
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
GSplatPrim::~GSplatPrim()
{
    if (!id_.empty()) renderer_.flushEntriesForMatchingDetail(id_);
}

void GSplatPrim::update(const void* detail, const GSplatCacheVersion& version, int64_t vtxOffset, const gsplat_attrs& a,
                        const float* barycenter)
{
    const int64_t n = (a.count > 0 && a.P) ? a.count : 0;
    count_ = n;
    missing_ = 0;
    if (n == 0) return;   // an empty primitive registers nothing (:211-216)

    P_.assign(a.P, a.P + 3 * n);
    Cd_.resize(3 * n);
    scale_.resize(3 * n);
    orient_.resize(4 * n);
    alpha_.resize(n);

    if (a.Cd) gsplat_quantize_half(a.Cd, Cd_.data(), 3 * n);
    else { missing_ |= GSPLAT_MISSING_CD; complain("point attribute Cd is missing: splats are drawn black"); Cd_.assign(3 * n, 0); }

    const float* op = a.Alpha ? a.Alpha : a.opacity;   // both present: Alpha is the one that counts
    if (op) std::memcpy(alpha_.data(), op, sizeof(float) * n);
    else { missing_ |= GSPLAT_MISSING_OPACITY; complain("neither opacity nor Alpha found: splats are drawn opaque"); alpha_.assign(n, 1.0f); }

    if (a.scale) gsplat_quantize_half(a.scale, scale_.data(), 3 * n);
    else { missing_ |= GSPLAT_MISSING_SCALE; complain("point attribute scale is missing: unit scale assumed"); scale_.assign(3 * n, 0x3c00); }

    if (a.orient) gsplat_quantize_half(a.orient, orient_.data(), 4 * n);
    else {
        missing_ |= GSPLAT_MISSING_ORIENT;
        complain("point attribute orient is missing: identity rotation assumed");
        for (int64_t i = 0; i < n; ++i) { orient_[4 * i] = orient_[4 * i + 1] = orient_[4 * i + 2] = 0; orient_[4 * i + 3] = 0x3c00; }
    }

    // spherical harmonics: the array attribute, else sh1.., else f_rest_0..
    const bool by_array = a.sh_coefficients && a.sh_coefficients_len >= 3;
    const bool by_vec3 = !by_array && a.sh && a.sh[0];
    const bool by_rest = !by_array && !by_vec3 && a.f_rest && a.f_rest[0];
    if (by_array || by_vec3 || by_rest) {
        shx_.resize(16 * n); shy_.resize(16 * n); shz_.resize(16 * n);
        if (by_array) {
            gsplat_pack_sh_from_array(a.sh_coefficients, n, a.sh_coefficients_len / 3, shx_.data(), shy_.data(), shz_.data());
        } else if (by_vec3) {
            const float* v[15];
            bool open = true;
            for (int j = 0; j < 15; ++j) { open = open && a.sh[j]; v[j] = open ? a.sh[j] : nullptr; }   // stops at the first gap
            gsplat_pack_sh_from_vec3(v, n, shx_.data(), shy_.data(), shz_.data());
        } else {
            const float* v[45];
            bool open = true;
            for (int j = 0; j < 45; ++j) { open = open && a.f_rest[j]; v[j] = open ? a.f_rest[j] : nullptr; }
            gsplat_pack_sh_from_frest(v, n, shx_.data(), shy_.data(), shz_.data());
        }
    } else {
        shx_.clear(); shy_.clear(); shz_.clear();
        missing_ |= GSPLAT_MISSING_SH;
        std::fprintf(stderr, "gsplat-hip warning: no spherical harmonics (looked for sh_coefficients, sh1..sh15, f_rest_0..44): flat colour\n");
    }

    float bc[3] = {0.0f, 0.0f, 0.0f};
    if (barycenter) std::memcpy(bc, barycenter, sizeof(bc));
    else {   // float32 running sum, then one division -- as GEO_PrimGsplat::baryCenter does
        for (int64_t i = 0; i < n; ++i) { bc[0] += P_[3 * i]; bc[1] += P_[3 * i + 1]; bc[2] += P_[3 * i + 2]; }
        for (int k = 0; k < 3; ++k) bc[k] /= static_cast<float>(n);
    }

    id_ = renderer_.registerUpdate(detail, version, vtxOffset, n, bc, P_.data(), Cd_.data(), alpha_.data(), scale_.data(),
                                   orient_.data(), hasSh() ? shx_.data() : nullptr, hasSh() ? shy_.data() : nullptr,
                                   hasSh() ? shz_.data() : nullptr, hasSh() ? n : 0);

    has_eye_ = a.explicit_camera_pos != nullptr;
    if (has_eye_) std::memcpy(eye_, a.explicit_camera_pos, sizeof(eye_));
    sh_order_ = 3;
    if (a.sh_order) {
        sh_order_ = *a.sh_order;
        if (sh_order_ < 0 || sh_order_ > 3) {
            std::fprintf(stderr, "gsplat-hip error: gsplat__sh_order = %d, but only 0..3 exist: spherical harmonics switched off\n", sh_order_);
            sh_order_ = 0;
            missing_ |= GSPLAT_BAD_SH_ORDER;
        }
    }
}

void GSplatPrim::render(bool beautyMode)
{
    if (count_ == 0 || id_.empty()) return;
    renderer_.setRenderingEnabled(beautyMode);
    renderer_.includeInRenderPass(id_);
    if (has_eye_) renderer_.setExplicitCameraPos(eye_);
    renderer_.setSphericalHarmonicsOrder(sh_order_);
}

// ---------------------------------------------------------------------------------------------
struct gsplat_prim {
    GSplatPrim* impl;
};

extern "C" GSplatRenderer* gsplat_renderer_impl(gsplat_renderer* h);

extern "C" {

gsplat_prim* gsplat_prim_create(gsplat_renderer* renderer)
{
    GSplatRenderer* r = gsplat_renderer_impl(renderer);
    if (!r) return nullptr;
    GSplatPrim* p = new (std::nothrow) GSplatPrim(*r);
    if (!p) return nullptr;
    gsplat_prim* h = new (std::nothrow) gsplat_prim{p};
    if (!h) delete p;
    return h;
}

void gsplat_prim_destroy(gsplat_prim* p)
{
    if (!p) return;
    delete p->impl;
    delete p;
}

int gsplat_prim_update(gsplat_prim* p, uint64_t detail, const int64_t version[4], int64_t vtx_offset,
                       const gsplat_attrs* attrs, const float* barycenter_or_null, char* id_out, int id_cap)
{
    if (!p || !version || !attrs) return -1;
    GSplatCacheVersion v;
    for (int k = 0; k < 4; ++k) v.e[k] = version[k];
    p->impl->update(reinterpret_cast<const void*>(static_cast<uintptr_t>(detail)), v, vtx_offset, *attrs, barycenter_or_null);
    const std::string& id = p->impl->registryId();
    if (id_out && id_cap > 0) {
        std::strncpy(id_out, id.c_str(), static_cast<size_t>(id_cap) - 1);
        id_out[id_cap - 1] = '\0';
    }
    return static_cast<int>(id.size());
}

void gsplat_prim_render(gsplat_prim* p, int beauty_mode) { if (p) p->impl->render(beauty_mode != 0); }
unsigned gsplat_prim_missing(gsplat_prim* p) { return p ? p->impl->missing() : 0u; }
int gsplat_prim_sh_order(gsplat_prim* p) { return p ? p->impl->shOrder() : 0; }
int gsplat_prim_has_sh(gsplat_prim* p) { return (p && p->impl->hasSh()) ? 1 : 0; }

const void* gsplat_prim_array(gsplat_prim* p, int what)
{
    if (!p || p->impl->count() == 0) return nullptr;
    const GSplatPrim& g = *p->impl;
    switch (what) {
    case 0: return g.P();
    case 1: return g.Cd();
    case 2: return g.alpha();
    case 3: return g.scale();
    case 4: return g.orient();
    case 5: return g.hasSh() ? g.shx() : nullptr;
    case 6: return g.hasSh() ? g.shy() : nullptr;
    case 7: return g.hasSh() ? g.shz() : nullptr;
    default: return nullptr;
    }
}

}  // extern "C"

"""CPU tests of the GSplatRenderer host shim (dry instance: registry / staging logic only).

They mirror how the reference drives its renderer (src/GR_GSplat.C:423-436,472-492 and
src/DM_GSplatHook.C:30-39) and assert the semantics of src/GSplatRenderer.C:141-153 (active-set
diff), :218-320 (registry), :336-376 (cap), :403-418 (origin), :551-563 (camera position) and
:660-678 (postRender)."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture()
def R(pkg):
    r = pkg.GSplatRenderer(-1)
    yield r
    r.close()


def _ctx(pkg, cam):
    return pkg.GSplatRenderer.context(cam)


def test_registry_ids_and_version_purge(pkg, R):
    a = pkg.scenes.make_scene(10, seed=1, sh=False)
    id1 = R.registerUpdate(0xABC0, (1, 2, 3, 4), 7, a)
    assert id1 == "0xabc0__7__1_2_3_4"                       # hex(gdp)__vtxoffset__cacheversion (:241-243)
    assert R.query(R.Q_REGISTRY_SIZE) == 1
    id2 = R.registerUpdate(0xABC0, (1, 2, 3, 5), 7, a)       # same detail, new cache version: old entry purged
    assert id2 != id1 and R.query(R.Q_REGISTRY_SIZE) == 1
    R.registerUpdate(0xDEF0, (1, 0, 0, 0), 0, a)
    R.registerUpdate(0xDEF0, (1, 0, 0, 0), 64, a)            # same detail+version, another primitive
    assert R.query(R.Q_REGISTRY_SIZE) == 3
    R.flushEntriesForMatchingDetail("0xdef0__0__1_0_0_0")    # ~GR_PrimGsplat: all entries of that detail go
    assert R.query(R.Q_REGISTRY_SIZE) == 1
    R.flushEntriesForMatchingDetail("no-such-id")
    assert R.query(R.Q_REGISTRY_SIZE) == 1


def test_frame_protocol_staging_and_ages(pkg, R):
    cam = pkg.camera.make_camera(64, 48)
    a = pkg.scenes.make_scene(100, seed=2, sh=True)
    b = pkg.scenes.make_scene(50, seed=3, sh=True)
    ia = R.registerUpdate(1, (1, 0, 0, 0), 0, a)
    ib = R.registerUpdate(2, (1, 0, 0, 0), 0, b)
    r = _ctx(pkg, cam)
    # nothing marked active -> nothing staged, nothing rendered
    R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert R.query(R.Q_STAGING_COUNT) == 0 and R.query(R.Q_RENDER_COUNT) == 0 and R.query(R.Q_CAN_RENDER) == 0
    assert R.query(R.Q_ENTRY_AGE, ia) == 0 and R.query(R.Q_ENTRY_AGE_SINCE_ACTIVE, ia) == -1
    # frame with both active
    R.includeInRenderPass(ia); R.includeInRenderPass(ib)
    R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert R.query(R.Q_STAGING_COUNT) == 1 and R.query(R.Q_RENDER_COUNT) == 1
    assert R.query(R.Q_SPLAT_COUNT) == 150 and R.query(R.Q_ACTIVE_STAGED) == 2 and R.query(R.Q_SH_PRESENT) == 1
    assert np.array_equal(R.origin(), (a.barycenter() + b.barycenter()) / np.float32(2))   # mean of barycentres
    assert R.query(R.Q_ENTRY_AGE_SINCE_ACTIVE, ia) == 0
    # same active set: no restaging
    R.includeInRenderPass(ia); R.includeInRenderPass(ib)
    R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert R.query(R.Q_STAGING_COUNT) == 1 and R.query(R.Q_RENDER_COUNT) == 2
    # only A: restaged; B ages since last active
    R.includeInRenderPass(ia)
    R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert R.query(R.Q_STAGING_COUNT) == 2 and R.query(R.Q_SPLAT_COUNT) == 100
    assert R.query(R.Q_ENTRY_AGE_SINCE_ACTIVE, ib) == 1 and R.query(R.Q_ENTRY_AGE, ib) == 3
    # postRender cleared the active flags: a frame without includeInRenderPass renders nothing
    R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert R.query(R.Q_RENDER_COUNT) == 3
    # setRenderingEnabled(false) (non-beauty viewport modes, src/GR_GSplat.C:472)
    R.setRenderingEnabled(False)
    R.includeInRenderPass(ia)
    R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert R.query(R.Q_RENDER_COUNT) == 3


def test_camera_position_and_explicit_override(pkg, R):
    cam = pkg.camera.make_camera(64, 48, frame=11)
    a = pkg.scenes.make_scene(10, seed=2, sh=False)
    ia = R.registerUpdate(1, (1, 0, 0, 0), 0, a)
    r = _ctx(pkg, cam)
    R.includeInRenderPass(ia)
    R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert np.allclose(R.lastCameraPos(), cam.cam_pos, atol=1e-6)          # translation of inverse(view)
    R.includeInRenderPass(ia)
    R.setExplicitCameraPos((1.5, -2.0, 0.25))                              # gsplat__explicit_camera_pos
    R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert np.array_equal(R.lastCameraPos(), np.float32([1.5, -2.0, 0.25]))
    R.includeInRenderPass(ia)                                              # cleared by postRender (:677)
    R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert np.allclose(R.lastCameraPos(), cam.cam_pos, atol=1e-6)


def test_sh_presence_follows_the_last_active_entry(pkg, R):
    cam = pkg.camera.make_camera(64, 48)
    with_sh = pkg.scenes.make_scene(20, seed=5, sh=True)
    without = pkg.scenes.make_scene(20, seed=6, sh=False)
    i1 = R.registerUpdate(0x10, (1, 0, 0, 0), 0, with_sh)
    i2 = R.registerUpdate(0x20, (1, 0, 0, 0), 0, without)
    r = _ctx(pkg, cam)
    R.includeInRenderPass(i1); R.includeInRenderPass(i2)
    R.generateRenderGeometry(r)
    last = max(i1, i2)                                                     # registry iteration order = id order
    assert R.query(R.Q_SH_PRESENT) == (1 if last == i1 else 0)
    R.postRender()


def test_splat_budget_cap(pkg):
    """2^23 - 1 splats are rendered at most; the excess is culled with a warning (:336-376).
    Dry instance + NULL arrays: only counts matter."""
    L = pkg.load_library()
    h = L.gsplat_renderer_create(-1)
    ver = (C.c_int64 * 4)(1, 0, 0, 0)
    org = (C.c_float * 3)(0, 0, 0)
    ids = []
    for k, cnt in enumerate((5_000_000, 3_000_000, 2_000_000, 1_000_000)):
        buf = C.create_string_buffer(128)
        L.gsplat_renderer_register_update(h, 0x100 + k, ver, 0, cnt, org, *([None] * 8), 0, buf, 128)
        ids.append(buf.value)
    for i in ids:
        L.gsplat_renderer_include_in_render_pass(h, i)
    r = pkg.engine.GSplatRenderContext()
    L.gsplat_renderer_generate_render_geometry(h, C.byref(r))
    assert L.gsplat_renderer_query(h, 2, None) == (1 << 23) - 1            # Q_SPLAT_COUNT
    # the third entry crosses the budget (it is truncated while packing); the fourth never becomes active
    assert L.gsplat_renderer_query(h, 1, None) == 3
    L.gsplat_renderer_destroy(h)
    for n, want in ((0, 2), (1, 2), (10_000, 128), (40_000, 256), (1_000_000, 1024), (6_000_000, 4096),
                    (24_000_000, 8192)):
        assert L.gsplat_closest_sqrt_power_of_2(n) == want


def test_blank_redraw_keeps_the_resident_copy_and_flush_invalidates_it(pkg, R):
    """a redraw that shows nothing leaves what is resident alone (no re-upload when the primitive comes back);
    destroying the primitive (flushEntriesForMatchingDetail) invalidates it, so the same id is staged again"""
    cam = pkg.camera.make_camera(64, 48)
    a = pkg.scenes.make_scene(100, seed=2, sh=True)
    ia = R.registerUpdate(7, (1, 0, 0, 0), 0, a)
    r = _ctx(pkg, cam)
    R.includeInRenderPass(ia); R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert R.query(R.Q_STAGING_COUNT) == 1
    R.generateRenderGeometry(r); R.render(r); R.postRender()                 # nothing shown
    R.includeInRenderPass(ia); R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert R.query(R.Q_STAGING_COUNT) == 1 and R.query(R.Q_RENDER_COUNT) == 2
    R.flushEntriesForMatchingDetail(ia)
    assert R.query(R.Q_REGISTRY_SIZE) == 0
    assert R.registerUpdate(7, (1, 0, 0, 0), 0, a) == ia                     # a new primitive under the same id
    R.includeInRenderPass(ia); R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert R.query(R.Q_STAGING_COUNT) == 2


# ------------------------------------------------------------------------------------------------
# N1: GSplatPrim -- GR_PrimGsplat's attribute ingest (src/GR_GSplat.C:93-189,233-372,438-457) and redraw verbs (:459-492)
def _raw_attrs(n, seed):
    rng = np.random.default_rng(seed)
    return {"P": rng.normal(0, 0.3, (n, 3)).astype(np.float32), "Cd": rng.random((n, 3), dtype=np.float32),
            "opacity": rng.random(n, dtype=np.float32), "scale": np.exp(rng.uniform(-5, -3, (n, 3))).astype(np.float32),
            "orient": rng.normal(0, 1, (n, 4)).astype(np.float32)}


def _h(a):
    return np.asarray(a, np.float32).astype(np.float16).view(np.uint16)


def test_prim_ingest_precedence_defaults_and_quantisation(pkg, R):
    n = 41
    at = _raw_attrs(n, 1)
    at["Alpha"] = np.linspace(0, 1, n, dtype=np.float32)            # both present: Alpha wins (:240-257)
    P = pkg.GSplatPrim(R)
    pid = P.update(0x77, (3, 0, 0, 0), 5, at)
    assert pid == "0x77__5__3_0_0_0" and R.query(R.Q_REGISTRY_SIZE) == 1
    a = P.arrays(n)
    assert np.array_equal(a.P, at["P"]) and np.array_equal(a.alpha, at["Alpha"])
    assert np.array_equal(a.Cd, _h(at["Cd"])) and np.array_equal(a.scale, _h(at["scale"])) and np.array_equal(a.orient, _h(at["orient"]))
    assert P.missing == pkg.engine.MISSING_SH and not P.has_sh and P.sh_order == 3
    del at["Alpha"]
    P.update(0x77, (4, 0, 0, 0), 5, at)
    assert np.array_equal(P.arrays(n).alpha, at["opacity"]) and R.query(R.Q_REGISTRY_SIZE) == 1     # old version purged
    # everything but P missing: Cd 0, alpha 1, scale 1, orient (0,0,0,1) (:309-313)
    P.update(0x77, (5, 0, 0, 0), 5, {"P": at["P"]})
    a = P.arrays(n)
    assert (a.Cd == 0).all() and (a.alpha == 1).all() and (a.scale == 0x3C00).all()
    assert (a.orient[:, :3] == 0).all() and (a.orient[:, 3] == 0x3C00).all()
    e = pkg.engine
    assert P.missing == e.MISSING_CD | e.MISSING_OPACITY | e.MISSING_SCALE | e.MISSING_ORIENT | e.MISSING_SH
    P.close()
    assert R.query(R.Q_REGISTRY_SIZE) == 0                          # ~GR_PrimGsplat flushes (:63-70)


def test_prim_three_sh_schemes_agree_and_sh_order_rule(pkg, R):
    n = 23
    rng = np.random.default_rng(5)
    coef = rng.normal(0, 0.1, (n, 15, 3)).astype(np.float32)        # sh_k of point i = coef[i, k-1]
    base = _raw_attrs(n, 2)
    want = [np.zeros((n, 16), np.uint16) for _ in range(3)]
    for ch in range(3):
        want[ch][:, :15] = _h(coef[:, :, ch])
    schemes = {
        "array": {"sh_coefficients": coef.reshape(n, 45)},
        "vec3": {f"sh{k + 1}": np.ascontiguousarray(coef[:, k, :]) for k in range(15)},
        "f_rest": {f"f_rest_{k + 15 * ch}": np.ascontiguousarray(coef[:, k, ch]) for k in range(15) for ch in range(3)},
    }
    P = pkg.GSplatPrim(R)
    for name, extra in schemes.items():
        P.update(0x10, (1, 0, 0, 0), 0, {**base, **extra})
        a = P.arrays(n)
        assert P.has_sh and P.missing == 0, name
        for ch, arr in enumerate((a.shx, a.shy, a.shz)):
            assert np.array_equal(arr, want[ch]), name
    # the array attribute wins over sh1.., which win over f_rest_ (:145-189)
    P.update(0x10, (1, 0, 0, 0), 0, {**base, **schemes["f_rest"], "sh1": np.zeros((n, 3), np.float32)})
    a = P.arrays(n)
    assert (a.shx == 0).all()                                       # sh1 (zeros) found, sh2.. missing -> zeros
    # gsplat__sh_order: 0..3 accepted, anything else -> 0 (:444-457)
    for given, used in ((0, 0), (2, 2), (3, 3), (4, 0), (-1, 0)):
        P.update(0x10, (1, 0, 0, 0), 0, {**base, **schemes["vec3"], "gsplat__sh_order": given})
        assert P.sh_order == used
        assert bool(P.missing & pkg.engine.BAD_SH_ORDER) == (given not in (0, 1, 2, 3))
    P.close()


def test_prim_redraw_verbs(pkg, R):
    """GR_PrimGsplat::render (:459-492): enable/disable by render mode, include, explicit camera, SH order"""
    cam = pkg.camera.make_camera(64, 48)
    at = _raw_attrs(30, 3)
    at["gsplat__explicit_camera_pos"] = (0.5, 1.5, -2.5)
    at["gsplat__sh_order"] = 1
    P = pkg.GSplatPrim(R)
    P.update(0x20, (1, 0, 0, 0), 0, at)
    r = _ctx(pkg, cam)
    P.render(True)
    R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert R.query(R.Q_RENDER_COUNT) == 1 and R.query(R.Q_SPLAT_COUNT) == 30
    assert np.array_equal(R.lastCameraPos(), np.float32([0.5, 1.5, -2.5]))
    assert np.allclose(R.origin(), at["P"].astype(np.float32).mean(axis=0), atol=1e-6)      # baryCenter of the primitive
    P.render(False)                                                  # non-beauty mode: registered but not drawn
    R.generateRenderGeometry(r); R.render(r); R.postRender()
    assert R.query(R.Q_RENDER_COUNT) == 1
    P.close()

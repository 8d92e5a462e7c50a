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

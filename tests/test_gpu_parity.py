"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerance: north_star asks for <= 1e-3 per channel.  Because the kernels implement the same
float32 op-order contract as the oracle, every per-splat record must match BIT FOR BIT and
the image may differ only by the blend kernel's early-out (bound 2^-14 * max colour).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-3
REC_FIELDS = ("cx", "cy", "a1x", "a1y", "b1x", "b1y", "r", "g", "b", "la")


def _check_image(img, ref, tol=TOL):
    assert img.shape == ref.shape
    err = np.abs(img - ref)
    assert np.isfinite(img).all()
    assert err.max() <= tol, f"max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"
    return float(err.max())


@pytest.mark.parametrize("n,w,h,sh,order,frame", [
    (1, 64, 64, False, 0, 0),
    (300, 96, 80, True, 3, 0),
    (5000, 200, 120, True, 3, 2),
    (20000, 320, 240, True, 2, 1),
    (20000, 320, 240, True, 1, 5),
    (20000, 333, 250, False, 3, 7),      # SH order requested but no SH data -> order 0; odd size
    (100000, 640, 360, True, 3, 3),
])
def test_frame_matches_oracle(pkg, oracle, engine, n, w, h, sh, order, frame):
    splats = pkg.scenes.make_scene(n, seed=100 + n % 97, sh=sh)
    cam = pkg.camera.make_camera(w, h, sh_order=order, frame=frame)
    engine.upload(splats)
    img = engine.render(cam)
    cam_o = pkg.camera.make_camera(w, h, sh_order=order if sh else 0, frame=frame)
    ref = oracle.render(splats, cam_o)
    _check_image(img, ref)


def test_records_bit_exact(pkg, oracle, engine):
    splats = pkg.scenes.make_scene(50000, seed=11, sh=True)
    cam = pkg.camera.make_camera(640, 360, sh_order=3, frame=4)
    engine.upload(splats, origin=(0.25, -0.5, 0.125))
    engine.render(cam)
    dev = engine.debug_records(splats.n)
    ref = oracle.preprocess(splats, cam, origin=(0.25, -0.5, 0.125))
    vis = dev["visible"] == 1
    assert vis.sum() > 1000
    # sort keys of every splat that survived culling, bit exact (culled splats are dropped before the sort)
    assert np.array_equal(dev["key"][vis].view(np.uint32), ref["key"][vis].view(np.uint32))
    assert (ref["visible"][vis] == 1).all()        # device culls a superset of what the oracle culls
    for f in REC_FIELDS:
        a, b = dev[f][vis].view(np.uint32), ref[f][vis].view(np.uint32)
        assert np.array_equal(a, b), f"field {f}: {np.count_nonzero(a != b)} mismatches"
    # bbox half extents are not parity-relevant: the device shrinks them to where alpha can reach 1/255
    for f in ("hx", "hy"):
        assert (dev[f][vis] <= ref[f][vis]).all() and (dev[f][vis] > 0).all()


def test_depth_order_matches_oracle(pkg, oracle, engine):
    splats = pkg.scenes.make_scene(70001, seed=5, sh=False)
    splats.P[1000:1100] = splats.P[2000:2100]      # exact ties -> index order
    cam = pkg.camera.make_camera(256, 256, sh_order=0, frame=9)
    engine.upload(splats)
    engine.render(cam)
    dev = engine.debug_depth_order(splats.n)          # the splats that survived culling, nearest first
    # ties are broken by the storage order: the Morton order of the positions, which the oracle restates independently
    order0 = engine.debug_storage_order(splats.n)
    assert sorted(order0.tolist()) == list(range(splats.n)) and not np.array_equal(order0, np.arange(splats.n))
    assert np.array_equal(order0, oracle.storage_order(splats.P))
    ref = oracle.host_sort_only(splats.P, cam.cam_pos)  # all splats, (distance^2, storage position) ascending
    assert dev.shape[0] > 0.9 * splats.n
    keep = np.zeros(splats.n, bool)
    keep[dev] = True
    assert np.array_equal(dev, ref[keep[ref]])
    assert keep[1000:1100].sum() > 80 and keep[2000:2100].sum() > 80   # the exact ties are in play


@pytest.mark.parametrize("n,bits", [(0, 32), (1, 32), (63, 8), (4096, 32), (4097, 13), (250001, 32), (1 << 20, 16)])
def test_radix_sort_pairs(pkg, engine, n, bits):
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 1 << bits, n, dtype=np.uint64).astype(np.uint32)
    if n > 100:
        keys[: n // 3] = keys[n // 3: 2 * (n // 3)]  # many duplicates: stability matters
    vals = np.arange(n, dtype=np.uint32)
    k, v = engine.debug_sort_pairs(keys, vals, bits)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order])
    assert np.array_equal(v, vals[order])
    # the small-frame form (k_bucket_scatter + k_radix_local): 1024 buckets of width 2^shift from lo, filled with atomics, every
    # bucket sorted on its own and its ties put back into payload order -- the same order, ties included, WHATEVER the predicted
    # range, as long as no bucket outgrows its region of 8192 keys (then the call reports it; the pipeline falls back)
    if n > 0:
        span_bits = max(int(keys.max()).bit_length(), 1)
        for lo, shift in ((0, max(span_bits - 10, 0)), (int(keys.min()), 0), (0, min(span_bits, 31)), (int(keys.max()) // 2, max(span_bits - 13, 0)),
                          (int(keys.max()) // 3, max(span_bits - 11, 0))):
            d = np.where(keys > lo, (keys.astype(np.int64) - lo) >> shift, 0).clip(0, 1023)
            if np.bincount(d, minlength=1024).max() > 8192:
                with pytest.raises(pkg.GsrError):
                    engine.debug_sort_pairs(keys, vals, bits, local=(lo, shift))
                continue
            k2, v2 = engine.debug_sort_pairs(keys, vals, bits, local=(lo, shift))
            assert np.array_equal(k2, k) and np.array_equal(v2, v), (lo, shift)


@pytest.mark.parametrize("n,w,h,frame,big,shard,layout", [
    (30000, 400, 300, 0, 0, (0, 1), 0),
    (150000, 1920, 1080, -7, 0, (0, 1), 0),      # frame < 0: the camera INSIDE the cloud -- the nearest splats (the first blocks of the depth
    (150000, 1920, 1080, -8, 0, (2, 4), 1),      #   order) fill the screen: those blocks are binned by one workgroup per ROW of super-tiles
    (20000, 1280, 720, 1, 3000, (0, 1), 0),      # 240 super-tiles, thousands of splats covering dozens of them each
    (20000, 4096, 4096, 2, 500, (0, 1), 0),      # the maximum: 256 super-tiles of 16x16 tiles
    (25000, 640, 480, 3, 1500, (1, 3), 0),       # row shard: a super-tile lists a splat only through an OWNED tile row
    (25000, 1920, 1080, 4, 1500, (5, 8), 0),
    (25000, 640, 480, 3, 1500, (1, 3), 1),       # contiguous bands (K1 drops far splats before the covariance chain)
    (25000, 1920, 1080, 4, 1500, (5, 8), 1),
])
def test_super_tile_lists_are_depth_ordered_and_complete(pkg, oracle, engine, n, w, h, frame, big, shard, layout):
    """the counting-sort binning (k_bin_count / k_bin_place): every super-tile's list holds exactly the splats
    whose tile rect reaches it (through an owned tile row), in depth order"""
    splats = pkg.scenes.make_scene(n, seed=21 + frame, sh=False)
    if big:
        sc = pkg.scenes.f16bits(np.random.default_rng(frame).uniform(0.5, 4.0, size=(big, 3)))
        splats.scale[:big] = sc
    cam = pkg.camera.make_camera(w, h, sh_order=0, frame=frame) if frame >= 0 else pkg.camera.make_camera(w, h, sh_order=0, frame=-frame, distance=0.35)
    engine.upload(splats)
    engine.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
    engine.set_row_shard(*shard)
    try:
        engine.render(cam)
        ls, le, pv = engine.debug_tile_lists()
        st = engine.stats()
        dev = engine.debug_records(splats.n)
    finally:
        engine.set_row_shard(0, 1)
        engine.set_option(pkg.engine.OPT_SHARD_LAYOUT, 0)
    S, sx = st["super_tile"], st["stiles_x"]
    assert ls.shape[0] == st["stiles_x"] * st["stiles_y"] <= 256 and st["pairs_total"] == pv.shape[0]
    rec = oracle.preprocess(splats, cam)
    perm = oracle.argsort(rec, oracle.storage_order(splats.P))
    rank = np.empty(splats.n, np.int64)
    rank[perm] = np.arange(splats.n)
    # expected membership from the (device-identical) pixel bbox -> tile rect
    vis = np.flatnonzero(dev["visible"] == 1)
    r = dev[vis]
    i0 = np.ceil(np.maximum(r["cx"] - r["hx"] - 0.5, 0)); i1 = np.floor(np.minimum(r["cx"] + r["hx"] - 0.5, w - 1))
    j0 = np.ceil(np.maximum(r["cy"] - r["hy"] - 0.5, 0)); j1 = np.floor(np.minimum(r["cy"] + r["hy"] - 0.5, h - 1))
    ok = (i1 >= i0) & (j1 >= j0)
    tx0, tx1, ty0, ty1 = (i0 // 16).astype(np.int64), (i1 // 16).astype(np.int64), (j0 // 16).astype(np.int64), (j1 // 16).astype(np.int64)
    idx, cnt = shard
    owned_rows = np.zeros((h + 15) // 16 + 1, np.int64)
    owned_rows[pkg.multigpu.owned_tile_rows(h, idx, cnt, layout)] = 1
    owned_before = np.concatenate([[0], np.cumsum(owned_rows)])      # owned rows among [0, r)
    seen = 0
    multi = 0
    for t in range(ls.shape[0]):
        lst = pv[ls[t]:le[t]]
        seen += lst.shape[0]
        if lst.shape[0] > 1:
            assert (np.diff(rank[lst]) > 0).all(), f"super-tile {t} not in depth order"
        X0, Y0 = (t % sx) * S, (t // sx) * S
        X1, Y1 = X0 + S - 1, Y0 + S - 1
        lo, hi = np.maximum(ty0, Y0), np.minimum(ty1, Y1)          # tile rows of the rect inside this super-tile
        some_owned = (hi >= lo) & (owned_before[np.clip(hi + 1, 0, len(owned_before) - 1)] - owned_before[np.clip(lo, 0, len(owned_before) - 1)] > 0)
        hit = ok & (tx1 >= X0) & (tx0 <= X1) & some_owned
        assert set(lst.tolist()) == set(vis[hit].tolist()), f"super-tile {t}: membership differs"
    assert seen == pv.shape[0]
    if big or frame < 0:
        cover = ((tx1 // S - tx0 // S + 1) * (ty1 // S - ty0 // S + 1))[ok]
        assert (cover > 8).sum() > 50       # the cooperative big-splat path was exercised
    if frame < 0:
        cov_near = cover[np.argsort(rank[vis[ok]])[:256]]          # the 256 nearest splats that draw are big on screen: dozens of super-tiles each
        assert np.median(cov_near) > 8 and cov_near.max() > 0.5 * ls.shape[0], (np.median(cov_near), cov_near.max())


@pytest.mark.parametrize("w,h,n,big", [
    (8192, 272, 30000, 2000),       # 512 tiles wide: tile rects are packed in pairs of tiles (GsrFrame.rect_shift = 1)
    (272, 8192, 30000, 2000),       # ... and 512 tiles high
    (4608, 4368, 40000, 1500),      # both sides beyond 4096: 288 x 273 tiles
    (16384, 272, 30000, 2000),      # 1024 tiles wide: rects in blocks of four tiles (rect_shift = 2)
    (272, 16384, 30000, 2000),      # ... and 1024 tiles high
])
def test_framebuffers_beyond_4096_pixels(pkg, oracle, engine, w, h, n, big):
    """the reference has no framebuffer limit (src/GSplatRenderer.C:534-657); here tile coordinates are packed in 8 bits, and a
    frame of more than 256 (512) tiles a side carries its rects in units of two (four) tiles -- a superset everywhere it is read,
    so the pixels are the oracle's, with and without occlusion culling, sharded or not"""
    splats = pkg.scenes.make_scene(n, seed=1234 + w, sh=True)
    splats.scale[:big] = pkg.scenes.f16bits(np.random.default_rng(w).uniform(0.3, 2.5, size=(big, 3)))
    cam = pkg.camera.make_camera(w, h, sh_order=3, frame=3)
    engine.upload(splats)
    img = engine.render(cam)
    st = engine.stats()
    assert st["stiles_x"] * st["stiles_y"] <= 256
    ref = oracle.render(splats, cam, threads=oracle.max_threads())
    assert ref[..., 3].max() > 0.5
    _check_image(img, ref)
    # every list in depth order, and holding at least the splats whose exact tile rect reaches the super-tile
    ls, le, pv = engine.debug_tile_lists()
    dev = engine.debug_records(splats.n)
    rec = oracle.preprocess(splats, cam)
    perm = oracle.argsort(rec, oracle.storage_order(splats.P))
    rank = np.empty(splats.n, np.int64)
    rank[perm] = np.arange(splats.n)
    vis = np.flatnonzero(dev["visible"] == 1)
    r = dev[vis]
    i0 = np.ceil(np.maximum(r["cx"] - r["hx"] - 0.5, 0)); i1 = np.floor(np.minimum(r["cx"] + r["hx"] - 0.5, w - 1))
    j0 = np.ceil(np.maximum(r["cy"] - r["hy"] - 0.5, 0)); j1 = np.floor(np.minimum(r["cy"] + r["hy"] - 0.5, h - 1))
    ok = (i1 >= i0) & (j1 >= j0)
    tx0, tx1, ty0, ty1 = (i0 // 16).astype(np.int64), (i1 // 16).astype(np.int64), (j0 // 16).astype(np.int64), (j1 // 16).astype(np.int64)
    S, sx = st["super_tile"], st["stiles_x"]
    extra = 0
    for t in range(ls.shape[0]):
        lst = pv[ls[t]:le[t]]
        if lst.shape[0] > 1:
            assert (np.diff(rank[lst]) > 0).all(), f"super-tile {t} not in depth order"
        X0, Y0 = (t % sx) * S, (t // sx) * S
        need = ok & (tx1 >= X0) & (tx0 <= X0 + S - 1) & (ty1 >= Y0) & (ty0 <= Y0 + S - 1)
        # ... and at most those whose rect, rounded outwards to the rect unit (pairs of tiles; blocks of four beyond 8192 pixels), does
        u = 3 if max(w, h) > 8192 else 1
        may = ok & ((tx1 | u) >= X0) & ((tx0 & ~u) <= X0 + S - 1) & ((ty1 | u) >= Y0) & ((ty0 & ~u) <= Y0 + S - 1)
        got = set(lst.tolist())
        assert set(vis[need].tolist()) <= got <= set(vis[may].tolist()), f"super-tile {t}: membership"
        extra += len(got) - int(need.sum())
    assert extra >= 0
    _same_with_culling(pkg, engine, cam, img)
    # row shards (both layouts): the owned tile rows of the full frame, bit for bit
    for layout, (idx, cnt) in ((0, (1, 3)), (1, (2, 4))):
        engine.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
        engine.set_row_shard(idx, cnt)
        try:
            band = engine.render(cam)
        finally:
            engine.set_row_shard(0, 1)
            engine.set_option(pkg.engine.OPT_SHARD_LAYOUT, 0)
        assert np.array_equal(band, pkg.multigpu.extract_band(img, idx, cnt, layout)), (layout, idx, cnt)


def test_the_largest_framebuffer(pkg, oracle, engine):
    """16384 x 16384 (GSR_MAX_DIM): 1024 x 1024 tiles, 256 super-tiles of 64 x 64 tiles, a 4.3 GB frame; three bands of rows against the oracle"""
    import ctypes as C
    w = h = pkg.engine.MAX_DIM
    assert w == 16384
    splats = pkg.scenes.make_scene(20000, seed=8192, sh=False)
    splats.scale[:1000] = pkg.scenes.f16bits(np.random.default_rng(8).uniform(0.3, 2.0, size=(1000, 3)))
    cam = pkg.camera.make_camera(w, h, sh_order=0, frame=1)
    engine.upload(splats)
    hip = C.CDLL("libamdhip64.so")
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(w * h * 16)) == 0
    try:
        engine.render_to_device(cam, p.value)
        engine.synchronize()
        st = engine.stats()
        assert st["super_tile"] == 64 and st["stiles_x"] * st["stiles_y"] == 256
        for y0 in (0, 8000, h - 96):
            got = np.empty((96, w, 4), np.float32)
            assert hip.hipMemcpy(C.c_void_p(got.ctypes.data), C.c_void_p(p.value + y0 * w * 16), C.c_size_t(got.nbytes), 2) == 0
            ref = oracle.render_rows(splats, cam, y0, y0 + 96, threads=oracle.max_threads())
            assert ref[..., 3].max() > 0.2
            _check_image(got, ref)
            del ref
        # the same frame with the heaviest-first tile order and occlusion culling forced on: bit-identical
        first = np.empty((256, w, 4), np.float32)
        assert hip.hipMemcpy(C.c_void_p(first.ctypes.data), C.c_void_p(p.value + 7900 * w * 16), C.c_size_t(first.nbytes), 2) == 0
        engine.set_option(pkg.engine.OPT_XCD_SWIZZLE, 3)
        engine.set_option(pkg.engine.OPT_OCCLUSION_CULL, 2)
        try:
            for _ in range(3):
                engine.render_to_device(cam, p.value)
            engine.synchronize()
            assert engine.stats()["frames_culled"] >= 1
        finally:
            engine.set_option(pkg.engine.OPT_XCD_SWIZZLE, 2)
            engine.set_option(pkg.engine.OPT_OCCLUSION_CULL, 1)
        again = np.empty_like(first)
        assert hip.hipMemcpy(C.c_void_p(again.ctypes.data), C.c_void_p(p.value + 7900 * w * 16), C.c_size_t(again.nbytes), 2) == 0
        assert np.array_equal(first, again)
    finally:
        hip.hipFree(p)
    with pytest.raises(pkg.GsrError):
        engine.render_to_device(pkg.camera.make_camera(w + 1, 64, sh_order=0), 0)


def _same_with_culling(pkg, engine, cam, img):
    """the same camera again with occlusion culling forced on (it engages from the slot's second frame): same pixels"""
    engine.set_option(pkg.engine.OPT_OCCLUSION_CULL, 2)
    try:
        for _ in range(3):
            assert np.array_equal(engine.render(cam), img, equal_nan=True), "the culled frame differs from the unculled one"
    finally:
        engine.set_option(pkg.engine.OPT_OCCLUSION_CULL, 1)


def test_edge_cases(pkg, oracle, engine):
    # empty cloud
    empty = pkg.scenes.make_scene(0, seed=1, sh=False)
    cam = pkg.camera.make_camera(64, 48, sh_order=0)
    engine.upload(empty)
    assert np.count_nonzero(engine.render(cam)) == 0
    _same_with_culling(pkg, engine, cam, np.zeros((48, 64, 4), np.float32))
    # everything behind the camera
    s = pkg.scenes.make_scene(1000, seed=2, sh=False)
    s.P[:] += np.float32(20.0) * cam.cam_pos / np.linalg.norm(cam.cam_pos)
    engine.upload(s)
    assert np.count_nonzero(engine.render(cam)) == 0
    # giant splats (axis cap) + opaque front splat hides what is behind it (blend state)
    s = pkg.scenes.make_scene(200, seed=3, sh=False)
    s.scale[:50] = pkg.scenes.f16bits(np.full((50, 3), 8.0))
    s.alpha[:] = 1.0
    engine.upload(s)
    img = engine.render(cam)
    ref = oracle.render(s, cam)
    _check_image(img, ref)
    assert img[..., 3].max() <= 1.0 + 1e-6
    _same_with_culling(pkg, engine, cam, img)


@pytest.mark.parametrize("layout", [0, 1])
def test_row_shards_stitch_to_the_same_image(pkg, engine, layout):
    """interleaved tile rows (layout 0) and contiguous bands (layout 1, with K1's early ownership test)"""
    splats = pkg.scenes.make_scene(40000, seed=31, sh=True)
    splats.scale[:300] = pkg.scenes.f16bits(np.random.default_rng(3).uniform(0.2, 2.0, size=(300, 3)))   # some span many rows
    cam = pkg.camera.make_camera(300, 200, sh_order=3, frame=2)
    engine.upload(splats)
    full = engine.render(cam)
    engine.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
    try:
        for count in (2, 3, 8):
            bands = []
            for idx in range(count):
                engine.set_row_shard(idx, count)
                band = engine.render(cam)
                assert band.shape[0] == engine.band_rows(cam.height)
                bands.append(band)
            engine.set_row_shard(0, 1)
            out = pkg.multigpu.stitch_bands_host(np.stack(bands), cam.height, layout)
            assert np.array_equal(out, full), f"layout {layout}, shard count {count}: stitched image differs"
    finally:
        engine.set_row_shard(0, 1)
        engine.set_option(pkg.engine.OPT_SHARD_LAYOUT, 0)


def test_renderer_shim_frame_protocol(pkg, oracle):
    R = pkg.GSplatRenderer(0)
    a = pkg.scenes.make_scene(3000, seed=41, sh=True)
    b = pkg.scenes.make_scene(2000, seed=42, sh=True)
    b.P[:] += np.float32([0.5, 0.2, 0.0])
    cam = pkg.camera.make_camera(256, 192, sh_order=3, frame=0)
    ida = R.registerUpdate(0x1000, (1, 0, 0, 0), 0, a)
    idb = R.registerUpdate(0x2000, (1, 0, 0, 0), 0, b)
    R.setSphericalHarmonicsOrder(3)
    img = R.frame(cam, [ida, idb])
    assert R.query(R.Q_STAGING_COUNT) == 1 and R.query(R.Q_RENDER_COUNT) == 1
    # oracle on the concatenation in registry (id) order with the mean-of-barycentres origin
    parts = [a, b] if ida < idb else [b, a]
    cat = pkg.scenes.Splats(*[np.concatenate([getattr(p, f) for p in parts]) for f in
                              ("P", "Cd", "alpha", "scale", "orient", "shx", "shy", "shz")])
    origin = (a.barycenter() + b.barycenter()) / np.float32(2)
    assert np.array_equal(R.origin(), origin)
    ref = oracle.render(cat, cam, origin=origin)
    _check_image(img, ref)
    # same active set -> no restaging; only A active -> restage
    R.frame(cam, [ida, idb])
    assert R.query(R.Q_STAGING_COUNT) == 1
    img_a = R.frame(cam, [ida])
    assert R.query(R.Q_STAGING_COUNT) == 2
    _check_image(img_a, oracle.render(a, cam, origin=a.barycenter()))
    R.close()


# ---------------------------------------------------------------------------------------------
# golden fixtures: the reference's own GLSL on a software rasteriser (tests/golden/make_goldens.py)
from helpers import (HipBuffers, check_against_golden, check_wire_against_golden, engine_render_golden, golden_names, golden_uncertainty,  # noqa: E402
                     load_golden, oracle_render_golden)


@pytest.mark.parametrize("name", golden_names())
def test_golden_reference_glsl_images(pkg, oracle, engine, name):
    d, s, c = load_golden(name)
    engine.upload(s, origin=d["origin"])
    img = engine_render_golden(engine, d, c)                   # (g8: depth-tested against the fixture's depth buffer)
    # vs the reference GLSL: every pixel beyond the 1e-3 budget sits on a quad edge / at a discard threshold (helpers.py)
    check_against_golden(img, d["image_reference_glsl"], uncertainty=golden_uncertainty(oracle, d, s, c), extra_tol=2.0 ** -14)
    _check_image(img, oracle_render_golden(oracle, d, s, c))   # vs the oracle: strict 1e-3 on every pixel
    # vertex stage vs the captured reference vertex shader outputs
    dev = engine.debug_records(s.n)
    vs = d["vs_out"]
    vis = dev["visible"] == 1
    if vis.any():
        assert np.abs(vs[vis, 0, 8:11] - np.stack([dev["r"][vis], dev["g"][vis], dev["b"][vis]], 1)).max() <= 1e-6
    # the same frame again with occlusion culling forced on (it engages from the slot's second frame): same pixels
    engine.set_option(pkg.engine.OPT_OCCLUSION_CULL, 2)
    try:
        for _ in range(3):
            assert np.array_equal(engine_render_golden(engine, d, c), img), "the culled frame differs from the first, unculled one"
    finally:
        engine.set_option(pkg.engine.OPT_OCCLUSION_CULL, 1)


def test_baseline_config_c2_full_size(pkg, oracle, engine):
    """BASELINE configs[1]: 100k anisotropic splats, SH degree 3, 1280x720"""
    splats, cfg = pkg.scenes.make_config("C2")
    cam = pkg.camera.make_camera(cfg["width"], cfg["height"], sh_order=cfg["sh_order"], frame=0)
    engine.upload(splats)
    img = engine.render(cam)
    _check_image(img, oracle.render(splats, cam, threads=oracle.max_threads()))


def test_baseline_config_c3_through_the_scene_pipeline(pkg, oracle, engine, tmp_path):
    """BASELINE configs[2]: 1M splats "from the hip file", 1920x1080.  The capture is not shipped, so raw INRIA-style
    attributes go through the example scene's own pipeline: PLY import -> activations -> fpreal16 cast ->
    Cd overwritten with 0.5 grey -> gsplat__sh_order 3."""
    cfg = pkg.scenes.CONFIGS["C3"]
    raw = pkg.scenes.make_inria_raw(cfg["n"], cfg["seed"], radius=cfg["radius"])
    path = str(tmp_path / "c3.ply")
    pkg.scenes.write_inria_ply(path, raw)
    splats = pkg.ply.load_inria_ply(path, cd_override=(0.5, 0.5, 0.5))
    assert splats.n == 1_000_000 and splats.has_sh and (splats.Cd == 0x3800).all()
    cam = pkg.camera.make_camera(cfg["width"], cfg["height"], sh_order=cfg["sh_order"], frame=0)
    engine.upload(splats, origin=splats.barycenter())
    img = engine.render(cam)
    ref = oracle.render(splats, cam, origin=splats.barycenter(), threads=oracle.max_threads())
    _check_image(img, ref)
    st = engine.stats()
    assert st["n_visible"] > 900_000 and img[..., 3].max() > 0.99
    # the synthetic stand-in bench.py uses for C3 (make_config) as well
    s2, _ = pkg.scenes.make_config("C3")
    engine.upload(s2)
    _check_image(engine.render(cam), oracle.render(s2, cam, threads=oracle.max_threads()))


def test_baseline_config_c5_4k_eight_shards_device_stitch(pkg, oracle, engine):
    """BASELINE configs[4]: 6M splats, 3840x2160, tile rows over 8 GPUs.  The full frame against the oracle (<= 1e-3 on
    every channel of every pixel), then the 8 row shards rendered one after the other into DEVICE band buffers laid out
    as the gather delivers them and de-interleaved by gsr_stitch_bands (k_stitch_bands): bit-identical to the
    unsharded frame."""
    splats, cfg = pkg.scenes.make_config("C5")
    W, H = cfg["width"], cfg["height"]
    cam = pkg.camera.make_camera(W, H, sh_order=3, frame=11)
    engine.upload(splats)
    full = engine.render(cam)
    ref = oracle.render(splats, cam, threads=oracle.max_threads())
    _check_image(full, ref)
    del ref
    G = 8
    hb = HipBuffers()
    try:
        for layout in (0, 1):
            engine.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
            engine.set_row_shard(0, G)
            rows = engine.band_rows(H)
            gathered = hb.alloc(G * rows * W * 16)
            final = hb.alloc(H * W * 16)
            for g in range(G):
                engine.set_row_shard(g, G)
                assert engine.band_rows(H) == rows
                engine.render_to_device(cam, gathered + g * rows * W * 16)
            engine.set_row_shard(0, 1)
            engine.stitch_bands(gathered, G, W, H, final)
            engine.synchronize()
            got = hb.download(final, (H, W, 4))
            assert np.array_equal(got, full), f"layout {layout}"
            hb.free()
    finally:
        engine.set_row_shard(0, 1)
        engine.set_option(pkg.engine.OPT_SHARD_LAYOUT, 0)
        hb.free()


@pytest.mark.parametrize("w,h,count,layout", [(300, 200, 3, 0), (641, 367, 2, 0), (1920, 1080, 8, 0), (100, 16, 5, 0),
                                              (300, 200, 3, 1), (1920, 1080, 8, 1), (100, 16, 5, 1)])
def test_stitch_bands_kernel(pkg, engine, w, h, count, layout):
    """gsr_stitch_bands (k_stitch_bands) against the host restatement of the de-interleave, on random bands"""
    rng = np.random.default_rng(w + h + count)
    rows = pkg.multigpu.band_rows(h, count)
    bands = rng.random((count, rows, w, 4), dtype=np.float32)
    hb = HipBuffers()
    try:
        src = hb.upload(bands)
        dst = hb.alloc(h * w * 16)
        engine.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
        engine.stitch_bands(src, count, w, h, dst)
        engine.synchronize()
        got = hb.download(dst, (h, w, 4))
    finally:
        engine.set_option(pkg.engine.OPT_SHARD_LAYOUT, 0)
        hb.free()
    assert np.array_equal(got, pkg.multigpu.stitch_bands_host(bands, h, layout))


def test_sort_cache_and_rotation_only_camera(pkg, oracle, engine):
    """The reference re-sorts only when the camera POSITION changes (src/GSplatRenderer.C:165-186).  Here the cached
    order holds just the splats visible to the frame that sorted, so it is reused for an identical frame only; a pure
    rotation about the eye re-sorts (and must match the oracle, whose order depends on the position alone)."""
    splats = pkg.scenes.make_scene(30000, seed=51, sh=True)
    cam = pkg.camera.make_camera(320, 200, sh_order=3, frame=1)
    engine.upload(splats)
    a = engine.render(cam)
    b = engine.render(cam)
    assert np.array_equal(a, b)
    # rotate the view about the eye: V' = R * V keeps cam_pos
    ang = 0.2
    rot = np.array([[np.cos(ang), 0, np.sin(ang), 0], [0, 1, 0, 0], [-np.sin(ang), 0, np.cos(ang), 0], [0, 0, 0, 1]])
    v = cam.view.reshape(4, 4).T.astype(np.float64)
    v2 = rot @ v
    cam2 = pkg.camera.Camera(obj_view=np.ascontiguousarray(v2.T, np.float32).reshape(16), object=cam.object,
                             inv_object=cam.inv_object, view=np.ascontiguousarray(v2.T, np.float32).reshape(16),
                             proj=cam.proj, cam_pos=cam.cam_pos, width=cam.width, height=cam.height, sh_order=3)
    img = engine.render(cam2)
    _check_image(img, oracle.render(splats, cam2))
    engine.set_option(pkg.engine.OPT_SORT_CACHE, 0)
    assert np.array_equal(img, engine.render(cam2))          # forced re-sort gives the same pixels
    engine.set_option(pkg.engine.OPT_SORT_CACHE, 1)


def test_options_do_not_change_pixels(pkg, engine):
    splats = pkg.scenes.make_scene(60000, seed=61, sh=True)
    cam = pkg.camera.make_camera(640, 400, sh_order=3, frame=3)
    engine.upload(splats)
    base = engine.render(cam)
    for opt, vals in ((pkg.engine.OPT_XCD_SWIZZLE, (0, 1, 3, 3, 2)), (pkg.engine.OPT_SUPER_TILE, (1, 2, 4, 8, 16, 0)),
                      (pkg.engine.OPT_DEBUG_FLAGS, (1, 2, 4, 7, 8, 15, 0)), (pkg.engine.OPT_FRAMES_IN_FLIGHT, (1, 2)),
                      (pkg.engine.OPT_LAZY_COLOUR, (0, 2, 1))):
        for v in vals:
            engine.set_option(opt, v)
            assert np.array_equal(engine.render(cam), base), f"option {opt}={v} changed the image"


@pytest.mark.gpu
@pytest.mark.parametrize("shard", [(0, 1, 0), (1, 3, 0), (2, 4, 1)])
def test_heaviest_first_tile_order_is_invisible(pkg, shard):
    """GSR_OPT_XCD_SWIZZLE = 3: from the second frame on the blend kernel's workgroup -> tile table comes from k_tile_order
    (the previous frame's per-tile work, heaviest first, per XCD).  Every tile must still be composited exactly once: a moving
    camera, so a tile left out would show the previous frame's pixels."""
    idx, count, layout = shard
    eng = pkg.Engine(0)
    splats = pkg.scenes.make_scene(50000, seed=67, sh=True)
    splats.scale[:200] = pkg.scenes.f16bits(np.random.default_rng(5).uniform(0.2, 1.5, size=(200, 3)))
    eng.upload(splats)
    eng.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
    eng.set_row_shard(idx, count)
    for w, h in ((1000, 700), (300, 2100)):
        cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=f) for f in range(5)]
        eng.set_option(pkg.engine.OPT_XCD_SWIZZLE, 0)
        want = [eng.render(c).copy() for c in cams]
        eng.set_option(pkg.engine.OPT_XCD_SWIZZLE, 3)
        for f, c in enumerate(cams):
            assert np.array_equal(eng.render(c), want[f]), f"{w}x{h} shard {shard}: frame {f} differs with the heaviest-first order"


def test_frames_in_flight_keep_every_frame_intact(pkg, oracle, engine):
    """two frames in flight: back-to-back asynchronous frames with different cameras into different
    device buffers must each equal the synchronous render of the same camera"""
    import ctypes as C
    L = pkg.load_library()
    splats = pkg.scenes.make_scene(200000, seed=71, sh=True)
    engine.upload(splats)
    w, h = 640, 360
    cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in range(6)]
    want = [engine.render(c) for c in cams]
    hip = C.CDLL("libamdhip64.so")
    bufs = []
    for _ in cams:
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(w * h * 16)) == 0
        bufs.append(p)
    for c, p in zip(cams, bufs):
        engine.render_to_device(c, p.value)            # no synchronisation between frames
    engine.synchronize()
    for k, p in enumerate(bufs):
        got = np.empty((h, w, 4), np.float32)
        assert hip.hipMemcpy(C.c_void_p(got.ctypes.data), p, C.c_size_t(w * h * 16), 2) == 0
        assert np.array_equal(got, want[k]), f"frame {k} differs"
        hip.hipFree(p)


def test_baseline_config_c4_full_size_properties(pkg, oracle, engine):
    """BASELINE's headline scene (6M splats, 1920x1080): idempotence, shard-stitch identity and,
    since the oracle finishes it in seconds on the box's cores, direct parity."""
    splats, cfg = pkg.scenes.make_config("C4")
    cam = pkg.camera.make_camera(cfg["width"], cfg["height"], sh_order=3, frame=7)
    engine.upload(splats)
    full = engine.render(cam)
    assert np.array_equal(full, engine.render(cam))
    st = engine.stats()
    assert st["n_visible"] > 1_000_000 and st["pairs_total"] >= st["n_visible"]
    tiles_y = (cam.height + 15) // 16
    out = np.zeros_like(full)
    for idx in range(4):
        engine.set_row_shard(idx, 4)
        band = engine.render(cam)
        for lrow, trow in enumerate(range(idx, tiles_y, 4)):
            y0, y1 = trow * 16, min(trow * 16 + 16, cam.height)
            out[y0:y1] = band[lrow * 16: lrow * 16 + (y1 - y0)]
    engine.set_row_shard(0, 1)
    assert np.array_equal(out, full)
    ref = oracle.render(splats, cam, threads=oracle.max_threads())
    _check_image(full, ref)
    # ... and against the reference's own GLSL program on this very scene: the band of tile rows held by the fixture
    import os
    from helpers import GOLDEN_DIR
    d = np.load(os.path.join(GOLDEN_DIR, "c4_band_1080p.npz"))
    y0, y1 = [int(v) for v in d["rows"]]
    img0 = engine.render(pkg.camera.make_camera(cfg["width"], cfg["height"], sh_order=3, frame=int(d["frame"])))
    # (every pixel of this band saturates: the kernel's early-out leaves each up to 2^-14 short, a one-sided bias the
    #  oracle -- which has no early-out -- does not show)
    check_against_golden(img0[y0:y1], d["band_reference_glsl"], max_bias=2.0 ** -14)


def test_adversarial_inputs(pkg, oracle, engine):
    """NaN/inf positions, zero / huge scales, negative and >1 opacities, degenerate quaternions"""
    rng = np.random.default_rng(5)
    s = pkg.scenes.make_scene(4000, seed=81, sh=True)
    s.P[10] = np.nan
    s.P[11, 0] = np.inf
    s.P[12] = -np.inf
    s.scale[20:40] = 0                                            # zero scale: only the 0.3 low-pass is left
    s.scale[40:46] = pkg.scenes.f16bits(np.full((6, 3), 3.0e4))   # far beyond the 4096-px axis cap
    s.scale[46:50] = pkg.scenes.f16bits(np.full((4, 3), np.inf))
    s.alpha[60:70] = -0.5
    s.alpha[70:80] = 37.0                                         # clamp(alpha, 0, 1)
    s.alpha[80:85] = np.nan
    s.alpha[85:90] = 1.0 / 255.0                                  # exactly at the discard threshold
    s.orient[90:100] = 0                                          # q = 0: R = I (no normalisation in the reference)
    s.orient[100:110] = pkg.scenes.f16bits(rng.normal(0, 30, (10, 4)))   # far from unit length
    s.shx[110:120] = 0x7C00                                       # +inf SH coefficients
    for frame, size in ((0, (320, 200)), (3, (257, 129))):
        cam = pkg.camera.make_camera(size[0], size[1], sh_order=3, frame=frame)
        engine.upload(s)
        img = engine.render(cam)
        ref = oracle.render(s, cam)
        ok = np.isfinite(ref)
        assert np.array_equal(np.isfinite(img), ok)              # non-finite colours propagate identically
        err = np.abs(img[ok] - ref[ok])
        assert err.max() <= TOL, err.max()
        _same_with_culling(pkg, engine, cam, img)


def test_maximum_size_through_the_shim(pkg, oracle):
    """2^23-1 splats is the reference's per-frame budget (include/GSplatRenderer.h:26): two entries that
    together exceed it are truncated while packing (src/GSplatRenderer.C:336-376,436-446)"""
    R = pkg.GSplatRenderer(0)
    a = pkg.scenes.make_scene(5_000_000, seed=91, sh=False, radius=2.0)
    b = pkg.scenes.make_scene(4_000_000, seed=92, sh=False, radius=2.0)
    cam = pkg.camera.make_camera(480, 270, sh_order=0, frame=2)
    ida = R.registerUpdate(0x10, (1, 0, 0, 0), 0, a)
    idb = R.registerUpdate(0x20, (1, 0, 0, 0), 0, b)
    img = R.frame(cam, [ida, idb])
    cap = (1 << 23) - 1
    assert R.query(R.Q_SPLAT_COUNT) == cap
    first, second = (a, b) if ida < idb else (b, a)
    keep = cap - first.n
    cat = pkg.scenes.Splats(*[np.concatenate([getattr(first, f), getattr(second, f)[:keep]]) for f in
                              ("P", "Cd", "alpha", "scale", "orient")])
    origin = (a.barycenter() + b.barycenter()) / np.float32(2)
    # the shim derives the camera position itself (inverse of the view matrix in double, as the reference
    # does, src/GSplatRenderer.C:556-562); with 8.4 M splats the sort has many near-ties, so the oracle
    # must be given exactly that position rather than numpy's last-ulp-different inverse
    cam.cam_pos = R.lastCameraPos()
    ref = oracle.render(cat, cam, origin=origin, threads=oracle.max_threads())
    _check_image(img, ref)
    R.close()


def test_depth_tested_compositing(pkg, oracle, engine):
    """splats behind what the opaque pass left in the depth buffer are rejected
    (depth test on, depth writes off -- src/GSplatRenderer.C:595-610)"""
    splats = pkg.scenes.make_scene(60000, seed=95, sh=True)
    cam = pkg.camera.make_camera(480, 300, sh_order=3, frame=4)
    rec = oracle.preprocess(splats, cam)
    zmid = float(np.median(rec["zwin"][rec["visible"] == 1]))
    yy, xx = np.mgrid[0:cam.height, 0:cam.width]
    depth = np.full((cam.height, cam.width), 1.0, np.float32)
    depth[(xx // 40 + yy // 40) % 2 == 0] = zmid            # checkerboard "wall" through the middle of the cloud
    depth[:20] = 0.0                                         # a strip that hides everything
    engine.upload(splats)
    img = engine.render_depth(cam, depth)
    ref = oracle.render_depth(splats, cam, depth)
    _check_image(img, ref)
    assert np.count_nonzero(img[:20]) == 0
    free = engine.render(cam)
    assert np.array_equal(engine.render_depth(cam, np.ones_like(depth)), free)     # depth = far plane: nothing rejected
    assert not np.array_equal(img, free)
    # sharded, with the full-size depth image
    out = np.zeros_like(img)
    tiles_y = (cam.height + 15) // 16
    for idx in range(2):
        engine.set_row_shard(idx, 2)
        band = engine.render_depth(cam, depth)
        for lrow, trow in enumerate(range(idx, tiles_y, 2)):
            y0, y1 = trow * 16, min(trow * 16 + 16, cam.height)
            out[y0:y1] = band[lrow * 16: lrow * 16 + (y1 - y0)]
    engine.set_row_shard(0, 1)
    assert np.array_equal(out, img)


def test_wireframe_overlay(pkg, oracle, engine):
    """quad outlines, colour Cd, nearest line wins -- bit-identical to the oracle, and within the
    line-rule slack of the reference's wire program on SwiftShader"""
    d, s, c = load_golden("w1_wire")
    engine.upload(s)
    img = engine.render_wire(c)
    assert np.array_equal(img, oracle.render_wire(s, c))
    check_wire_against_golden(img, d["wire_reference_glsl"])
    big = pkg.scenes.make_scene(200000, seed=97, sh=True)
    big.scale[:20] = pkg.scenes.f16bits(np.full((20, 3), 50.0))      # axis cap: outlines far larger than the screen
    cam = pkg.camera.make_camera(640, 400, sh_order=3, frame=6, distance=1.2)   # some splats behind the eye
    engine.upload(big)
    assert np.array_equal(engine.render_wire(cam), oracle.render_wire(big, cam))


def test_wire_over_keeps_the_beauty_frame(pkg, oracle, engine):
    """Wire-over display (the reference draws the outlines and still includes the primitive in the splat pass,
    src/GR_GSplat.C:471-486): gsr_render_wire_over writes the pixels an outline covers and leaves the beauty frame
    everywhere else -- host target and device target alike"""
    d, s, c = load_golden("w1_wire")
    engine.upload(s)
    beauty = engine.render(c)
    wire = engine.render_wire(c)
    both = engine.render_wire_over(c, beauty)
    covered = wire[..., 3] > 0
    assert covered.any() and (~covered).any()
    assert np.array_equal(both[covered], wire[covered])
    assert np.array_equal(both[~covered], beauty[~covered])
    hb = HipBuffers()                        # (a device target: raw hipMalloc, not torch -- torch's own HIP runtime does not come up behind ours)
    try:
        ptr = hb.upload(beauty)
        engine.render_wire_over_device(c, ptr)
        assert np.array_equal(hb.download(ptr, beauty.shape), both)
    finally:
        hb.free()


def test_wire_overlay_after_a_failed_upload_uses_the_new_cloud(pkg, oracle):
    """The geometry generation only counts up: after an upload that failed (an announced count that was never filled) the next
    cloud must not be drawn with the inverse storage permutation of the one before it (round-4 advisor finding: the generation was
    rewound to 0, so the second cloud was generation 1 again)."""
    eng = pkg.Engine(0)
    try:
        a = pkg.scenes.make_scene(5000, seed=11, sh=True)
        cam = pkg.camera.make_camera(320, 200, sh_order=3, frame=3)
        eng.upload(a)
        assert np.array_equal(eng.render_wire(cam), oracle.render_wire(a, cam))
        L = eng.L
        assert L.gsr_upload_begin(eng.h, 1000, 0, None) == 0
        assert L.gsr_upload_end(eng.h) != 0          # nothing was appended: the upload fails ...
        L.gsr_upload_abort(eng.h)
        b = pkg.scenes.make_scene(1200, seed=12, sh=True)    # ... and a smaller cloud follows
        eng.upload(b)
        assert np.array_equal(eng.render_wire(cam), oracle.render_wire(b, cam))
        assert np.abs(eng.render(cam) - oracle.render(b, cam)).max() <= 1e-3
    finally:
        eng.close()


def test_list_buffer_regrows_behind_a_speculative_back_end(pkg, oracle):
    """The back end (placement + compositing) is queued before the host knows the frame's pair count,
    clamped to the list buffer's capacity; when the count exceeds it the buffer is regrown and the back
    end re-run before gsr_render returns.  A frame with few pairs followed by frames with many more
    (two successively bigger scenes) must still be exact."""
    eng = pkg.Engine(0)
    try:
        for fif in (1, 2):
            eng.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, fif)
            small = pkg.scenes.make_scene(500, seed=5, sh=False)
            eng.upload(small)
            eng.render(pkg.camera.make_camera(128, 96, sh_order=0, frame=0))
            d0 = eng.stats()["pairs_total"]
            big = pkg.scenes.make_scene(60000, seed=6, sh=True)
            eng.upload(big)
            cam = pkg.camera.make_camera(800, 600, sh_order=3, frame=2)
            img = eng.render(cam)
            d1 = eng.stats()["pairs_total"]
            assert d1 > 4 * max(d0, 1)
            _check_image(img, oracle.render(big, cam))
            assert np.array_equal(img, eng.render(cam))      # now non-speculative growth is over: same pixels
            huge = pkg.scenes.make_scene(300000, seed=7, sh=True)            # 5x the splats on 4.3x the pixels
            eng.upload(huge)
            cam2 = pkg.camera.make_camera(1920, 1080, sh_order=3, frame=2)
            img2 = eng.render(cam2)
            assert eng.stats()["pairs_total"] > 2 * d1
            _check_image(img2, oracle.render(huge, cam2, threads=oracle.max_threads()))
    finally:
        eng.close()


@pytest.mark.parametrize("seed", range(10))
def test_random_frames_match_oracle(pkg, oracle, engine, seed):
    """seeded random frame descriptions (size, SH order, camera, splat sizes, super-tile edge, frames in flight)"""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 40000))
    w, h = int(rng.integers(17, 900)), int(rng.integers(17, 700))
    sh = bool(rng.integers(0, 2))
    order = int(rng.integers(0, 4))
    splats = pkg.scenes.make_scene(n, seed=2000 + seed, sh=sh)
    nbig = int(rng.integers(0, max(1, n // 10)))
    if nbig:
        splats.scale[:nbig] = pkg.scenes.f16bits(rng.uniform(0.3, 5.0, size=(nbig, 3)))
    frame = int(rng.integers(0, 120))
    cam = pkg.camera.make_camera(w, h, sh_order=order, frame=frame)
    engine.set_option(pkg.engine.OPT_SUPER_TILE, int(rng.choice([0, 1, 2, 4, 8, 16])))
    engine.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, int(rng.choice([1, 2])))
    try:
        engine.upload(splats)
        img = engine.render(cam)
    finally:
        engine.set_option(pkg.engine.OPT_SUPER_TILE, 0)
        engine.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, 2)
    cam_o = pkg.camera.make_camera(w, h, sh_order=order if sh else 0, frame=frame)   # (no SH data: the doSH gate forces order 0)
    _check_image(img, oracle.render(splats, cam_o, threads=oracle.max_threads()))


# ---------------------------------------------------------------------------------------------
# several GPUs from one thread (gsr_multi_*): on the 1-GPU box the ranks are contexts on the same GPU (transport COPY);
# shard, render, gather, stitch are the code the multi-GPU node runs, only the transport of the gather differs
@pytest.mark.parametrize("ranks,layout", [(2, 0), (3, 0), (8, 0), (3, 1), (8, 1)])
def test_multi_gpu_in_library_matches_single_gpu(pkg, oracle, engine, ranks, layout):
    splats = pkg.scenes.make_scene(60000, seed=131, sh=True)
    cam = pkg.camera.make_camera(500, 333, sh_order=3, frame=5)
    engine.upload(splats)
    want = engine.render(cam)
    with pkg.MultiEngine([0] * ranks, pkg.engine.TRANSPORT_COPY) as M:
        assert M.count == ranks and M.transport == pkg.engine.TRANSPORT_COPY
        M.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
        M.upload(splats)
        got = M.render(cam)
        assert np.array_equal(got, want)
        for f in range(3):                                         # an orbit: every frame re-sorts on every rank
            c = pkg.camera.make_camera(500, 333, sh_order=3, frame=20 + f)
            assert np.array_equal(M.render(c), engine.render(c))
        vis = [M.stats(r)["n_visible"] for r in range(ranks)]
        # each rank keeps only the splats of its rows (a rank of the band layout may own no row at all: 21 tile rows / 8)
        assert all(0 <= v <= engine.stats()["n_visible"] for v in vis) and sum(v > 0 for v in vis) >= ranks - 1
        # device target + device depth on the root
        hb = HipBuffers()
        try:
            rec = oracle.preprocess(splats, cam)
            zmid = float(np.median(rec["zwin"][rec["visible"] == 1]))
            depth = np.full((cam.height, cam.width), 1.0, np.float32)
            depth[:, : cam.width // 2] = zmid
            d_dev = hb.upload(depth)
            out = hb.alloc(cam.height * cam.width * 16)
            M.render_struct_to_device(pkg.engine.camera_struct(cam), out, d_dev)
            M.synchronize()
            got_d = hb.download(out, (cam.height, cam.width, 4))
        finally:
            hb.free()
        assert np.array_equal(got_d, engine.render_depth(cam, depth))
        assert np.array_equal(M.render(cam, depth), got_d)
    _check_image(want, oracle.render(splats, cam))


@pytest.mark.parametrize("direct,items", [("2", "1"), ("-1", "2"), ("0", "4"), ("2", "4")])
def test_small_frame_kernel_variants_do_not_change_pixels(pkg, engine, direct, items):
    """the small-frame sort's scatter (general gathering / one wave per K1 block / one atomic per key) and the binning kernels'
    splats per thread are chosen by frame size; forced through the A/B hooks, every combination renders the same frames"""
    import os
    splats = pkg.scenes.make_scene(300000, seed=77, sh=True)
    w, h = 800, 608
    cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in range(6)]
    engine.upload(splats)
    engine.set_option(pkg.engine.OPT_OCCLUSION_CULL, 2)
    try:
        want = [engine.render(c) for c in cams]
    finally:
        engine.set_option(pkg.engine.OPT_OCCLUSION_CULL, 1)
    old = {k: os.environ.get(k) for k in ("GSR_SCATTER_DIRECT", "GSR_BN_ITEMS")}
    os.environ["GSR_SCATTER_DIRECT"], os.environ["GSR_BN_ITEMS"] = direct, items
    try:
        eng = pkg.Engine(0)                     # (the hooks are read when a context is created)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    try:
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 2)
        eng.set_option(pkg.engine.OPT_LOCAL_SORT, 2)      # the small-frame sort whatever the frame keeps
        eng.upload(splats)
        for k, c in enumerate(cams):
            assert np.array_equal(eng.render(c), want[k]), f"frame {k} differs"
        assert eng.stats()["frames_culled"] >= 4 and eng.stats()["frames_resorted"] == 0
    finally:
        eng.close()


def test_multi_gpu_back_to_back_device_frames_with_uneven_ranks(pkg, engine):
    """COPY transport, device target, no synchronisation between frames, bands of very different weight (the ball fills the
    middle bands, the outer ones see sky): a light rank is frames ahead of the root's copies unless its stream waits for them
    -- every frame must still be the single-GPU frame, bit for bit"""
    splats, _ = pkg.scenes.make_config("B1", 400000)
    w, h = 960, 544
    cams = [pkg.scenes.config_camera("B1", pkg.camera, w, h, 3, i) for i in range(8)]
    engine.upload(splats)
    want = [engine.render(c) for c in cams]
    hb = HipBuffers()
    try:
        outs = [hb.alloc(w * h * 16) for _ in cams]
        with pkg.MultiEngine([0] * 4, pkg.engine.TRANSPORT_COPY) as M:
            M.set_option(pkg.engine.OPT_SHARD_LAYOUT, 1)
            M.upload(splats)
            for c, o in zip(cams, outs):
                M.render_struct_to_device(pkg.engine.camera_struct(c), o, 0)        # (no synchronisation in between)
            M.synchronize()
            vis = [M.stats(r)["n_visible"] for r in range(4)]
            assert max(vis) > 3 * max(1, min(vis))                                  # the ranks really are uneven
            for k, o in enumerate(outs):
                assert np.array_equal(hb.download(o, (h, w, 4)), want[k]), f"frame {k} differs"
    finally:
        hb.free()


def test_multi_gpu_behind_the_renderer_verbs(pkg, oracle):
    """GSplatRenderer over two contexts: the nine verbs drive the sharded path from the one draw thread"""
    a = pkg.scenes.make_scene(20000, seed=141, sh=True)
    cam = pkg.camera.make_camera(320, 240, sh_order=3, frame=1)
    R1 = pkg.GSplatRenderer(0)
    R2 = pkg.GSplatRenderer([0, 0], pkg.engine.TRANSPORT_COPY)
    try:
        i1 = R1.registerUpdate(0x1, (1, 0, 0, 0), 0, a)
        i2 = R2.registerUpdate(0x1, (1, 0, 0, 0), 0, a)
        img1, img2 = R1.frame(cam, [i1]), R2.frame(cam, [i2])
        assert R2.query(R2.Q_RENDER_COUNT) == 1 and np.array_equal(img1, img2)
        _check_image(img2, oracle.render(a, cam, origin=a.barycenter()))
    finally:
        R1.close(); R2.close()


def test_deferred_pair_count_check(pkg):
    """GSR_OPT_DEFERRED_CHECK: device-target frames return without the host reading the pair count; pixels stay exact as
    long as the list buffer (sized from earlier frames + 25 %) holds the frame -- and the library counts when it did not"""
    splats = pkg.scenes.make_scene(150000, seed=151, sh=True)
    w, h = 640, 400
    cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in range(8)]
    engine = pkg.Engine(0)      # its own context: the list buffer must be as small as this scene made it
    engine.upload(splats)
    want = [engine.render(c) for c in cams]
    hb = HipBuffers()
    try:
        bufs = [hb.alloc(w * h * 16) for _ in cams]
        engine.set_option(pkg.engine.OPT_DEFERRED_CHECK, 1)
        engine.stats_reset()
        for c, p in zip(cams, bufs):
            engine.render_to_device(c, p)
        engine.synchronize()
        st = engine.stats()
        assert st["frames"] == len(cams) and st["frames_truncated"] == 0 and st["frames_requeued"] == 0
        for k, p in enumerate(bufs):
            assert np.array_equal(hb.download(p, (h, w, 4)), want[k]), f"frame {k}"
        # a scene with far more pairs than the buffer was sized for: the deferred frame is clamped AND reported ...
        big = pkg.scenes.make_scene(150000, seed=152, sh=True, log_scale_range=(-3.5, -2.5))
        engine.upload(big)
        engine.render_to_device(cams[0], bufs[0])
        engine.synchronize()
        assert engine.stats()["frames_truncated"] == 1
        # ... and the next frame is exact again (buffer regrown)
        engine.render_to_device(cams[0], bufs[1])
        engine.synchronize()
        engine.set_option(pkg.engine.OPT_DEFERRED_CHECK, 0)
        assert np.array_equal(hb.download(bufs[1], (h, w, 4)), engine.render(cams[0]))
    finally:
        hb.free()
        engine.close()


def test_sort_cache_counts_hits(pkg, engine):
    """the exact-frame sort cache is per frame slot: with one slot the second identical frame skips the sort, a rotated
    camera (same position) and a moved camera both re-sort"""
    splats = pkg.scenes.make_scene(30000, seed=161, sh=False)
    cam = pkg.camera.make_camera(256, 160, sh_order=0, frame=1)
    engine.upload(splats)
    engine.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, 1)
    try:
        engine.stats_reset()
        a = engine.render(cam)
        assert engine.stats()["sorts_skipped"] == 0
        b = engine.render(cam)
        c = engine.render(cam)
        assert engine.stats()["sorts_skipped"] == 2 and np.array_equal(a, b) and np.array_equal(a, c)
        engine.render(pkg.camera.make_camera(256, 160, sh_order=0, frame=2))
        assert engine.stats()["sorts_skipped"] == 2
        engine.set_option(pkg.engine.OPT_SORT_CACHE, 0)
        assert np.array_equal(engine.render(cam), a) and engine.stats()["sorts_skipped"] == 2
    finally:
        engine.set_option(pkg.engine.OPT_SORT_CACHE, 1)
        engine.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, 2)


@pytest.mark.parametrize("cull", [0, 2])
def test_position_keyed_order_skips_the_sort_on_a_rotation(pkg, oracle, cull):
    """GSR_OPT_SORT_CACHE = 2, the reference's rule (src/GSplatRenderer.C:165-186: argsortByDistance runs only when the camera
    POSITION moves): while the position stands still the frames take the depth order of all splats sorted for that position --
    sorts_skipped counts them -- and render exactly what a context that sorts every frame renders"""
    splats = pkg.scenes.make_scene(120000, seed=163, sh=True)
    w, h = 480, 320
    base = pkg.camera.make_camera(w, h, sh_order=3, frame=2)
    turns = [pkg.camera.rotated_in_place(base, y, p) for y, p in ((0, 0), (4, 0), (8, 2), (-6, -3), (12, 1))]
    moved = pkg.camera.make_camera(w, h, sh_order=3, frame=9)
    ref = pkg.Engine(0); ref.upload(splats); ref.set_option(pkg.engine.OPT_OCCLUSION_CULL, cull)
    eng = pkg.Engine(0); eng.upload(splats); eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, cull)
    try:
        eng.set_option(pkg.engine.OPT_SORT_CACHE, 2)
        eng.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, 1)
        eng.stats_reset()
        assert np.array_equal(eng.render(turns[0]), ref.render(turns[0]))
        assert eng.stats()["sorts_skipped"] == 0                      # a position seen for the first time: sorted as ever
        for k, c in enumerate(turns[1:], start=1):
            assert np.array_equal(eng.render(c), ref.render(c)), f"turn {k}"
            # the position has not moved: the order stands (a frame that occlusion culling renders twice skips twice)
            assert eng.stats()["sorts_skipped"] - eng.stats()["frames_repaired"] == k
        before = eng.stats()["sorts_skipped"]
        assert np.array_equal(eng.render(moved), ref.render(moved))
        assert eng.stats()["sorts_skipped"] == before                   # ... and a move sorts again
        assert np.array_equal(eng.render(pkg.camera.rotated_in_place(moved, 5)), ref.render(pkg.camera.rotated_in_place(moved, 5)))
        assert eng.stats()["sorts_skipped"] > before
        _check_image(eng.render(turns[2]), oracle.render(splats, turns[2], threads=oracle.max_threads()))
    finally:
        eng.close(); ref.close()


@pytest.mark.parametrize("scheme", ["array", "vec3", "f_rest", "none"])
def test_prim_ingest_renders_like_the_oracle(pkg, oracle, scheme):
    """SURVEY N1 on the GPU: raw float attributes in each SH naming scheme -> GSplatPrim::update (quantise, pack, register)
    -> the redraw verbs -> pixels, against the oracle fed with numpy's own quantisation"""
    n = 30000
    s = pkg.scenes.make_scene(n, seed=171, sh=True)
    rng = np.random.default_rng(172)
    coef = rng.normal(0, 0.15, (n, 15, 3)).astype(np.float32)
    f16 = lambda bits: bits.view(np.float16).astype(np.float32)
    at = {"P": s.P, "Cd": f16(s.Cd), "opacity": s.alpha, "scale": f16(s.scale) * np.float32(1.0003),   # not fp16-exact
          "orient": f16(s.orient), "gsplat__sh_order": 2}
    if scheme == "array":
        at["sh_coefficients"] = coef.reshape(n, 45)
    elif scheme == "vec3":
        at.update({f"sh{k + 1}": np.ascontiguousarray(coef[:, k, :]) for k in range(15)})
    elif scheme == "f_rest":
        at.update({f"f_rest_{k + 15 * ch}": np.ascontiguousarray(coef[:, k, ch]) for k in range(15) for ch in range(3)})
    h = lambda a: np.asarray(a, np.float32).astype(np.float16).view(np.uint16)
    ref_s = pkg.scenes.Splats(s.P, s.Cd, s.alpha, h(at["scale"]), s.orient)
    if scheme != "none":
        sh = [np.zeros((n, 16), np.uint16) for _ in range(3)]
        for ch in range(3):
            sh[ch][:, :15] = h(coef[:, :, ch])
        ref_s.shx, ref_s.shy, ref_s.shz = sh
    cam = pkg.camera.make_camera(400, 300, sh_order=2 if scheme != "none" else 0, frame=3)
    R = pkg.GSplatRenderer(0)
    P = pkg.GSplatPrim(R)
    try:
        P.update(0x99, (1, 0, 0, 0), 0, at)
        out = np.zeros((cam.height, cam.width, 4), np.float32)
        r = R.context(cam, out.ctypes.data, False)
        P.render(True)
        R.generateRenderGeometry(r); R.render(r, False); R.postRender()
        assert R.query(R.Q_RENDER_COUNT) == 1
        cam.cam_pos = R.lastCameraPos()
        _check_image(out, oracle.render(ref_s, cam, origin=R.origin(), threads=oracle.max_threads()))
        assert out[..., 3].max() > 0.5
    finally:
        P.close(); R.close()


@pytest.mark.parametrize("shard", [(0, 1, 0), (1, 2, 1)])
def test_occlusion_culling_is_exact(pkg, oracle, shard):
    """GSR_OPT_OCCLUSION_CULL: from a slot's second frame on, splats behind the previous frame's depth horizons are dropped
    before projection, sorting and binning.  Every frame must equal the unculled render bit for bit: in a steady orbit
    (culling engaged, horizons hold), across a camera jump and a jump in distance (horizons break: the frame is rendered
    again), with a depth buffer, and with two frames in flight."""
    idx, count, layout = shard
    eng = pkg.Engine(0)
    try:
        splats = pkg.scenes.make_scene(400000, seed=191, sh=True, radius=1.0)
        w, h = 960, 540
        eng.upload(splats)
        eng.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
        eng.set_row_shard(idx, count)
        cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in (0, 1, 2, 3, 4, 40, 41, 42)]
        cams += [pkg.camera.make_camera(w, h, sh_order=3, frame=43, distance=d) for d in (2.2, 2.25, 6.0, 5.9)]
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
        want = [eng.render(c).copy() for c in cams]
        if count == 1:
            _check_image(want[0], oracle.render(splats, cams[0], threads=oracle.max_threads()))
        vis_full = eng.stats()["n_visible"]
        for fif in (1, 2):
            eng.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, fif)
            eng.set_option(pkg.engine.OPT_XCD_SWIZZLE, 3 if fif == 2 else 2)   # (with the heaviest-first tile order on top)
            eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 2)         # whenever a slot has horizons (1 = when it is found to pay)
            eng.upload(splats)                                    # forget the horizons
            eng.stats_reset()
            vis = []
            for k, (c, ref) in enumerate(zip(cams, want)):
                assert np.array_equal(eng.render(c), ref), f"frames in flight {fif}: frame {k} differs with occlusion culling"
                vis.append(eng.stats()["n_visible"])
            st = eng.stats()
            assert st["frames_culled"] >= len(cams) - 2 * fif - 1, st
            assert st["frames_repaired"] <= 4, st
            assert min(vis) < 0.8 * vis_full, (vis, vis_full)   # it does cull
        if count == 1:   # several GPUs' worth of contexts behind gsr_multi: every rank culls and checks its own band
            eng.set_row_shard(0, 1)
            M = pkg.MultiEngine([0, 0, 0], transport=pkg.engine.TRANSPORT_COPY)
            try:
                M.set_option(pkg.engine.OPT_SHARD_LAYOUT, 1)
                M.set_option(pkg.engine.OPT_OCCLUSION_CULL, 2)
                M.upload(splats)
                for k, (c, ref) in enumerate(zip(cams, want)):
                    assert np.array_equal(M.render(c), ref), f"gsr_multi: frame {k} differs with occlusion culling"
            finally:
                M.close()
            eng.set_row_shard(idx, count)
        # with the opaque pass's depth in front of part of the cloud fewer tiles go opaque: still exact
        rng = np.random.default_rng(7)
        depth = np.where(rng.random((h, w)) < 0.5, 0.2, 1.0).astype(np.float32)
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
        want_d = [eng.render_depth(c, depth).copy() for c in cams[:4]]
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 2)
        eng.upload(splats)
        for k, (c, ref) in enumerate(zip(cams[:4], want_d)):
            assert np.array_equal(eng.render_depth(c, depth), ref), f"depth-tested frame {k} differs with occlusion culling"
    finally:
        eng.close()


@pytest.mark.parametrize("seed", [3, 4])
def test_occlusion_culling_random_walk(pkg, seed):
    """a random walk of the camera (orbit steps, jumps, distance changes), of the resolution and of the engine's options, with
    occlusion culling forced on: every frame equals the same frame of an engine that never culls, bit for bit"""
    rng = np.random.default_rng(seed)
    splats = pkg.scenes.make_scene(300000, seed=200 + seed, sh=True, radius=1.0)
    ref, eng = pkg.Engine(0), pkg.Engine(0)
    try:
        ref.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 2)
        ref.upload(splats); eng.upload(splats)
        frame, dist, size = 0, 2.2, (800, 450)
        repaired = 0
        for step in range(40):
            u = rng.random()
            if u < 0.70: frame += 1                                    # a steady orbit step
            elif u < 0.80: frame += int(rng.integers(10, 60))           # a jump
            elif u < 0.90: dist = float(rng.choice([1.6, 2.2, 3.0, 5.0]))
            else: size = (800, 450) if size != (800, 450) else (512, 512)
            if rng.random() < 0.15:
                eng.set_option(pkg.engine.OPT_SUPER_TILE, int(rng.choice([0, 4, 8])))
            if rng.random() < 0.15:
                eng.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, int(rng.choice([1, 2])))
            if rng.random() < 0.10:
                eng.set_option(pkg.engine.OPT_XCD_SWIZZLE, int(rng.choice([1, 2, 3])))
            order = int(rng.choice([0, 3, 3, 3]))
            cam = pkg.camera.make_camera(size[0], size[1], sh_order=order, frame=frame, distance=dist)
            assert np.array_equal(eng.render(cam), ref.render(cam)), f"step {step}: frame {frame}, distance {dist}, {size}, SH {order}"
        st = eng.stats()
        assert st["frames_culled"] >= 10, st          # (culling was really exercised, and some horizons really broke)
    finally:
        ref.close(); eng.close()


def test_cluster_culling_and_storage_order_are_invisible(pkg, oracle):
    """k_cluster.h: the splats are stored in Morton order and every frame starts by culling clusters of 64 of them (clip
    planes, screen, band, depth horizons).  The cluster stage may not change a pixel: frames with it switched off
    (GSR_OPT_CLUSTER_CULL = 0) are bit-identical, sharded and with occlusion culling on top too, and without occlusion culling
    exactly the same splats reach the depth sort.  The storage order (GSR_OPT_STORAGE_ORDER) only decides the order of splats at
    exactly the same distance, as the oracle's tie order does: each order matches the oracle told the same."""
    splats = pkg.scenes.make_scene(300000, seed=77, sh=True, radius=1.0)
    w, h = 800, 450
    cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in (0, 1, 2, 3)]
    cams += [pkg.camera.make_camera(w, h, sh_order=3, frame=7, distance=d) for d in (0.3, 0.9, 1.6, 8.0)]   # inside the cloud, ..., far away
    cams += [pkg.camera.make_camera(w, h, sh_order=3, frame=9, distance=2.5, pivot=(1.5, 0.4, 0.0))]          # most of the cloud off screen
    # the projections of golden g9 / g10 / g11 (round 4): the cluster bounds divide by view z and take their clamp limits from P00
    # like the per-splat code, so they must stay conservative when clip w is 1 (orthographic; wide and narrow; the eye plane
    # cutting through the cloud), when the frustum is off-centre, and under a wide lens in portrait format
    cm = pkg.camera
    for hw, dist, fr in ((1.35, 4.62, 3), (0.5, 4.62, 4), (1.2, 0.7, 5)):
        cams.append(cm.make_camera(w, h, sh_order=3, frame=fr, distance=dist, proj_matrix=cm.orthographic(-hw, hw, -hw * h / w, hw * h / w, 0.05, 60.0)))
    a = 0.05 / 2.41421
    cams.append(cm.make_camera(w, h, sh_order=3, frame=7, proj_matrix=cm.frustum(-0.6 * a, 1.4 * a, -1.3 * a * h / w, 0.7 * a * h / w, 0.05, 1.0e4)))
    cams.append(cm.make_camera(w, h, sh_order=3, frame=8, distance=1.4, proj_matrix=cm.frustum(0.2 * a, 1.8 * a, 0.1 * a, 1.3 * a, 0.05, 1.0e4)))   # the axis off screen
    cams.append(cm.make_camera(w, h, sh_order=3, frame=11, p00=1.1, distance=2.6))
    n_plain = 9
    ref = pkg.Engine(0)
    eng = pkg.Engine(0)
    try:
        for storage in (1, 0):
            for e in (ref, eng):
                e.set_row_shard(0, 1)
                e.set_option(pkg.engine.OPT_STORAGE_ORDER, storage)
                e.upload(splats)
            ref.set_option(pkg.engine.OPT_CLUSTER_CULL, 0)
            ref.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
            order0 = eng.debug_storage_order(splats.n)
            oracle.set_tie_order(storage == 0)
            try:
                assert np.array_equal(order0, oracle.storage_order(splats.P))
                assert np.array_equal(order0, np.arange(splats.n)) == (storage == 0)
                want, vis = [], []
                for c in cams:
                    want.append(ref.render(c).copy())
                    vis.append(ref.stats()["n_visible"])
                _check_image(want[0], oracle.render(splats, cams[0], threads=oracle.max_threads()))
                if storage == 1:   # ... and the frames of the other projections are the oracle's frames
                    for k in range(n_plain, len(cams)):
                        _check_image(want[k], oracle.render(splats, cams[k], threads=oracle.max_threads()))
            finally:
                oracle.set_tie_order(False)
            for cull in (0, 2):
                eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, cull)
                for k, (c, img) in enumerate(zip(cams, want)):
                    assert np.array_equal(eng.render(c), img), f"storage {storage}, occlusion culling {cull}: frame {k} differs"
                    if cull == 0:
                        assert eng.stats()["n_visible"] == vis[k]
            # the cull kernel's several-rounds-per-workgroup form (clouds beyond 33 M splats), forced by a debug flag
            eng.set_option(pkg.engine.OPT_DEBUG_FLAGS, 16)
            for k, (c, img) in enumerate(zip(cams, want)):
                assert np.array_equal(eng.render(c), img), f"storage {storage}, three cull rounds per workgroup: frame {k} differs"
            eng.set_option(pkg.engine.OPT_DEBUG_FLAGS, 0)
            # a row shard on top (band layout: clusters outside the band are culled as a whole)
            for e in (ref, eng):
                e.set_option(pkg.engine.OPT_SHARD_LAYOUT, 1)
                e.set_row_shard(1, 3)
            for k, c in enumerate(cams[:5]):
                assert np.array_equal(eng.render(c), ref.render(c)), f"storage {storage}, band 1/3: frame {k} differs"
            for e in (ref, eng):
                e.set_option(pkg.engine.OPT_SHARD_LAYOUT, 0)
    finally:
        ref.close(); eng.close()


@pytest.mark.parametrize("kind", ["terrain", "slab"])
def test_occlusion_culling_on_scenes_with_sky_and_silhouettes(pkg, oracle, kind):
    """Per-tile depth horizons: a landscape under open sky (a third of the frame empty, the skyline crossing tile rows) and a
    thin wall whose silhouette sweeps across the frame as the camera orbits.  Forced on, culling must leave every frame bit-
    identical to the unculled one -- steady orbit, jumps, a depth-tested pass -- and under its own policy (the default) it must
    actually engage on such scenes: most frames culled, few repaired, far fewer splats through the depth sort."""
    w, h = 960, 540
    if kind == "terrain":
        splats = pkg.scenes.make_terrain(500000, seed=31, sh=True)
        cam_at = lambda i, **kw: pkg.scenes.terrain_camera(pkg.camera, w, h, frame=i, **kw)
    else:
        splats = pkg.scenes.make_slab(400000, seed=32, sh=True)
        cam_at = lambda i, **kw: pkg.camera.make_camera(w, h, sh_order=3, frame=i, distance=kw.get("distance", 5.5))
    frames = list(range(0, 12)) + [40, 41, 42, 75, 76, 77]          # a steady orbit, then two jumps
    cams = [cam_at(i) for i in frames] + [cam_at(78, distance=d) for d in (2.5, 2.6, 7.0)]
    ref, eng = pkg.Engine(0), pkg.Engine(0)
    try:
        ref.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
        ref.set_option(pkg.engine.OPT_CLUSTER_CULL, 0)
        ref.upload(splats); eng.upload(splats)
        want = [ref.render(c).copy() for c in cams]
        _check_image(want[0], oracle.render(splats, cams[0], threads=oracle.max_threads()))
        a = want[0][..., 3]
        assert (a < 0.01).mean() > 0.1 and (a > 0.99).mean() > 0.3        # empty sky AND opaque regions in one frame
        vis_full = ref.stats()["n_visible"]
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 2)
        vis = []
        for k, (c, img) in enumerate(zip(cams, want)):
            assert np.array_equal(eng.render(c), img), f"{kind}: frame {k} differs with occlusion culling"
            vis.append(eng.stats()["n_visible"])
        st = eng.stats()
        assert st["frames_culled"] >= len(cams) - 2 and st["frames_repaired"] <= 6, st
        # depth-tested frames (an opaque pass in front of half of the pixels)
        rng = np.random.default_rng(9)
        depth = np.where(rng.random((h, w)) < 0.5, 0.5, 1.0).astype(np.float32)
        want_d = [ref.render_depth(c, depth).copy() for c in cams[:5]]
        for k, (c, img) in enumerate(zip(cams[:5], want_d)):
            assert np.array_equal(eng.render_depth(c, depth), img), f"{kind}: depth-tested frame {k} differs with occlusion culling"
        # the library's own policy on a steady orbit of such a scene: exact either way; if it leaves culling alone, the reason is on record
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 1)
        eng.upload(splats)
        eng.stats_reset()
        orbit = [cam_at(i) for i in range(100, 140)]
        for k, c in enumerate(orbit):
            img = eng.render(c)
            if k % 5 == 0:
                assert np.array_equal(img, ref.render(c)), f"{kind}: orbit frame {k} differs under the default policy"
        st = eng.stats()
        assert st["frames_repaired"] <= 0.1 * len(orbit), st
        if st["frames_culled"] < 0.5 * len(orbit):
            assert vis_full < 300000 or not (st["policy_bits"] & 4) or (st["policy_bits"] & 8), st   # too small / nothing to gain / keeps > 70 %
    finally:
        ref.close(); eng.close()


def test_live_policy_state_follows_the_frames(pkg):
    """The policies of a LIVE context (csrc/gsr_policy.h; every transition is pinned on the CPU by tests/test_policy.py) move as the
    state table in DESIGN.md section 4 says when real frames drive them: a fresh cloud knows nothing; the first (unculled) frame sets the
    yardstick; once the kernels say culling pays the orbit is culled and the streak of frames that held counts up; a camera JUMP is
    recognised on the host (no culled attempt, no repair); the option is the floor of the dilation radius; a new cloud forgets."""
    E = pkg.engine
    splats = pkg.scenes.make_scene(600000, seed=43, sh=True, radius=1.0)
    w, h = 960, 540
    cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i, distance=3.2) for i in range(40)]
    eng = pkg.Engine(0)
    try:
        eng.set_option(E.OPT_CULL_DILATE, 3)
        eng.upload(splats)
        p = eng.policy_state()
        assert (p["cull_pays"], p["vis_unculled"], p["cull_holdoff"], p["cull_backoff"], p["cull_streak"], p["cull_dilate"], p["opt_dilate"]) == (0, 0, 0, 8, 0, 3, 3)
        eng.render(cams[0])
        p = eng.policy_state()
        assert p["vis_unculled"] == eng.stats()["n_visible"] > 100000 and eng.stats()["frames_culled"] == 0
        for c in cams[1:30]:
            eng.render(c)
        st, p = eng.stats(), eng.policy_state()
        assert p["cull_pays"] == 1 and st["frames_culled"] >= 20, (st, p)
        held = st["frames_culled"] - st["frames_repaired"]
        assert p["cull_streak"] == held % 64 if st["frames_repaired"] == 0 else p["cull_streak"] <= held
        assert p["cull_dilate"] >= 3 and (p["cull_dilate"] == 3 or st["frames_repaired"] > 0)
        before = eng.stats()
        eng.render(pkg.camera.make_camera(w, h, sh_order=3, frame=75, distance=4.4))       # a cut: 135 degrees round, 1.4 x as far
        after = eng.stats()
        assert after["frames_jumped"] == before["frames_jumped"] + 1 and after["frames_repaired"] == before["frames_repaired"]
        assert eng.policy_state()["cull_streak"] == p["cull_streak"]                        # (no verdict was asked for: the streak stands)
        eng.upload(pkg.scenes.make_scene(5000, seed=44, sh=False))
        p = eng.policy_state()
        assert (p["cull_pays"], p["vis_unculled"], p["cull_streak"], p["cull_dilate"], p["slab_holdoff"], p["local_fails"]) == (0, 0, 0, 3, 0, 0)
    finally:
        eng.close()


def test_occlusion_culling_engages_on_a_ball_under_open_sky(pkg):
    """A dense ball in the middle of an empty frame (a third of the pixels are sky, the limb crosses hundreds of tiles): with
    per-tile horizons the library's own policy culls (nearly) every frame of a steady orbit, repairs (nearly) none, sends a
    fraction of the splats through the depth sort -- and every frame is bit-identical to the unculled one."""
    splats = pkg.scenes.make_scene(1200000, seed=41, sh=True, radius=1.0)
    w, h = 1280, 720
    cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i, distance=3.6) for i in range(70)]
    ref, eng = pkg.Engine(0), pkg.Engine(0)
    try:
        ref.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
        ref.upload(splats); eng.upload(splats)
        a = ref.render(cams[0])[..., 3]
        vis_full = ref.stats()["n_visible"]
        assert (a < 0.01).mean() > 0.25 and (a > 0.99).mean() > 0.3
        for c in cams[:10]:
            eng.render(c)                       # the policy makes up its mind
        eng.stats_reset()
        vis = []
        for k, c in enumerate(cams[10:]):
            img = eng.render(c)
            vis.append(eng.stats()["n_visible"])
            if k % 6 == 0:
                assert np.array_equal(img, ref.render(c)), f"orbit frame {k} differs"
        st = eng.stats()
        assert st["frames_culled"] >= 0.8 * 60 and st["frames_repaired"] <= 0.05 * 60, st
        assert np.median(vis) < 0.5 * vis_full, (np.median(vis), vis_full)
    finally:
        ref.close(); eng.close()


def test_small_frame_sort_survives_mispredictions(pkg):
    """GSR_OPT_LOCAL_SORT (k_sort.h): a frame that keeps few splats is sorted by one global bucket pass over the key range the
    PREVIOUS frame kept + one local kernel.  The prediction only shapes the buckets: frames equal those of the three-pass sort
    bit for bit through steady orbits, camera jumps and jumps in distance (a bucket far beyond the prediction is given up and
    the frame rendered again with the global sort: frames_resorted), culled and unculled, static redraws included."""
    splats = pkg.scenes.make_scene(400000, seed=191, sh=True, radius=1.0)
    w, h = 960, 540
    cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in (0, 1, 2, 3, 4, 40, 41, 42)]
    cams += [pkg.camera.make_camera(w, h, sh_order=3, frame=43, distance=d) for d in (2.2, 2.25, 2.25, 6.0, 5.9, 5.9)]   # (with static redraws)
    ref = pkg.Engine(0)
    try:
        ref.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
        ref.set_option(pkg.engine.OPT_LOCAL_SORT, 0)
        ref.upload(splats)
        want = [ref.render(c).copy() for c in cams]
    finally:
        ref.close()
    for local, cull in ((1, 2), (1, 0), (2, 2), (2, 0)):
        eng = pkg.Engine(0)
        try:
            eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, cull)
            eng.set_option(pkg.engine.OPT_LOCAL_SORT, local)
            eng.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, 1)    # (one frame slot: a static redraw meets the order it left)
            eng.upload(splats)
            for k, (c, img) in enumerate(zip(cams, want)):
                assert np.array_equal(eng.render(c), img), f"local sort {local}, culling {cull}: frame {k} differs"
            st = eng.stats()
            assert st["frames_resorted"] >= 1, st          # the jumps in distance did overrun a bucket
            if cull == 0:
                assert st["sorts_skipped"] >= 2, st        # ... and the static redraws reused the (good) order
        finally:
            eng.close()


def test_lazy_colour_is_exact_and_predicts(pkg, oracle):
    """k_colour.h: SH colours are evaluated ahead of time only for the front of every super-tile list (as deep as the
    previous frame scanned); tiles that meet a pending colour fall back to on-demand evaluation.  Pixels equal eager
    evaluation bit for bit in every regime: first frame (everything coloured), steady orbit (prediction), camera jump
    and prediction switched off (fallback)."""
    eng = pkg.Engine(0)
    try:
        splats = pkg.scenes.make_scene(400000, seed=181, sh=True, radius=1.0)
        w, h = 960, 540
        eng.upload(splats)
        cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in (0, 1, 2, 3, 40, 41)]   # 3 deg steps, then a jump
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)      # (culled frames colour what K1 kept instead: their own test)
        eng.set_option(pkg.engine.OPT_LAZY_COLOUR, 0)
        want = [eng.render(c) for c in cams]
        _check_image(want[0], oracle.render(splats, cams[0], threads=oracle.max_threads()))
        eng.set_option(pkg.engine.OPT_LAZY_COLOUR, 2)         # always (1 = only when the kernels find that it pays)
        eng.upload(splats)                                    # forget the prediction
        eng.stats_reset()
        redo, colours = [], []
        for c, ref in zip(cams, want):
            assert np.array_equal(eng.render(c), ref)
            st = eng.stats()
            redo.append(st["lazy_redo_tiles"]); colours.append(st["lazy_colours_total"])
        per_frame = np.diff([0] + colours)
        nvis = eng.stats()["n_visible"]
        assert redo[0] == 0 and per_frame[0] >= 0.95 * nvis   # first frame: every listed splat is coloured ahead of time
        assert per_frame[2] < 0.8 * per_frame[0]              # steady state: a fraction of the visible splats (a small, shallow scene)
        assert max(redo[1:4]) <= 0.05 * ((w // 16) * (h // 16 + 1))   # ... and the prediction holds for (nearly) every tile
        # no ahead-of-time colours at all: every tile that composites anything goes through the fallback
        eng.set_option(pkg.engine.OPT_DEBUG_FLAGS, 8)
        assert np.array_equal(eng.render(cams[1]), want[1])
        assert eng.stats()["lazy_redo_tiles"] > 100
        eng.set_option(pkg.engine.OPT_DEBUG_FLAGS, 0)
        # records read back for tests are complete (pending colours evaluated on demand by the read-back)
        eng.render(cams[2])
        dev = eng.debug_records(splats.n)
        ref = oracle.preprocess(splats, cams[2])
        vis = dev["visible"] == 1
        for f in ("r", "g", "b"):
            assert np.array_equal(dev[f][vis].view(np.uint32), ref[f][vis].view(np.uint32))
    finally:
        eng.close()


def test_multi_gpu_over_rccl_matches_single_gpu(pkg, engine):
    """The frame's one collective between REAL GPUs: gsr_multi over min(visible, 8) devices with the RCCL transport (band images
    -> devices[0] by ncclSend / ncclRecv in one group, stitched there), occlusion culling on every rank, both shard layouts:
    bit-identical to the single-GPU frame.  Needs >= 2 GPUs (the 1-GPU box runs the same path over the COPY transport in
    test_multi_gpu_in_library_matches_single_gpu)."""
    ndev = int(pkg.load_library().gsr_device_count())
    if ndev < 2:
        pytest.skip("needs at least two GPUs")
    splats = pkg.scenes.make_scene(500000, seed=61, sh=True, radius=1.0)
    cams = [pkg.camera.make_camera(1280, 720, sh_order=3, frame=i) for i in (0, 1, 2, 3, 30, 31)]
    engine.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
    engine.upload(splats)
    try:
        want = [engine.render(c).copy() for c in cams]
    finally:
        engine.set_option(pkg.engine.OPT_OCCLUSION_CULL, 1)
    for layout in (1, 0):
        M = pkg.MultiEngine(list(range(min(ndev, 8))), transport=pkg.engine.TRANSPORT_RCCL)
        try:
            assert M.transport == pkg.engine.TRANSPORT_RCCL
            M.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
            M.set_option(pkg.engine.OPT_OCCLUSION_CULL, 2)
            M.upload(splats)
            for k, (c, img) in enumerate(zip(cams, want)):
                assert np.array_equal(M.render(c), img), f"layout {layout}: frame {k} differs between {M.count} GPUs and one"
        finally:
            M.close()


@pytest.mark.parametrize("scheme", ["array", "vec3", "f_rest", "none"])
def test_raw_ingest_on_the_gpu_matches_the_host_quantisation(pkg, scheme):
    """gsr_upload_append_raw: raw float32 attributes are quantised (fp16, round to nearest even, overflow to infinity) and packed
    on the GPU.  The frame -- records, keys, pixels -- is bit-identical to the one from GSplatPrim's host-side quantisation
    (src/GR_GSplat.C:302-372 restated) of the same attributes, in every SH naming scheme, with missing attributes taking the
    reference's defaults, and with values on every edge of the conversion: subnormal halves, the 65504 / 65520 rounding
    boundary, +-inf, NaN, -0, ties to even."""
    n = 40000
    s = pkg.scenes.make_scene(n, seed=271, sh=True)
    rng = np.random.default_rng(272)
    f16 = lambda bits: bits.view(np.float16).astype(np.float32)
    coef = rng.normal(0, 0.15, (n, 15, 3)).astype(np.float32)
    scale = f16(s.scale) * np.float32(1.0003)                              # not fp16-exact
    Cd = f16(s.Cd) + np.float32(3e-5)
    orient = f16(s.orient) * np.float32(0.99991)
    edge = np.float32([6.0e-8, 5.9604645e-8, 2.9802322e-8, 2.98023224e-8 * 1.0000001, 6.1e-5, 6.097e-5, 65504.0, 65519.99, 65520.0, 65536.0, 1e9,
                       np.inf, -np.inf, np.nan, -0.0, 0.0, 1.00048828125, 1.0009765625 + 0.00048828125, -2.0009765625 - 0.0009765625, 0.333333343])
    k = edge.size
    coef[:k, 3, 1] = edge; coef[k:2 * k, 0, 0] = -edge          # SH slots
    Cd[:k, 2] = edge                                            # colour
    scale[2 * k:3 * k, 1] = edge                                # scales (huge / inf / NaN ones drop their splat on both paths)
    orient[3 * k:4 * k, 0] = edge
    raw = {"P": s.P, "Cd": Cd, "alpha": s.alpha, "scale": scale, "orient": orient}
    if scheme == "array":
        raw["sh_coefficients"] = np.concatenate([coef.reshape(n, 45), rng.normal(0, 1, (n, 9)).astype(np.float32)], axis=1)   # 18 vec3: 16 are used
    elif scheme == "vec3":
        raw.update({f"sh{j + 1}": np.ascontiguousarray(coef[:, j, :]) for j in range(12)})                                      # sh13.. absent: zeros
    elif scheme == "f_rest":
        raw.update({f"f_rest_{j + 15 * ch}": np.ascontiguousarray(coef[:, j, ch]) for j in range(15) for ch in range(3)})
    # the host path: GSplatPrim's quantisers (parallel C++), then the half arrays through gsr_upload
    q = pkg.engine.quantize_half
    host = pkg.scenes.Splats(s.P, q(Cd), s.alpha, q(scale), q(orient))
    if scheme != "none":
        L = pkg.load_library()
        import ctypes as C
        sh = [np.zeros((n, 16), np.uint16) for _ in range(3)]
        if scheme == "array":
            arr = np.ascontiguousarray(raw["sh_coefficients"])
            L.gsplat_pack_sh_from_array(arr.ctypes.data, n, 18, sh[0].ctypes.data, sh[1].ctypes.data, sh[2].ctypes.data)
        elif scheme == "vec3":
            ptrs = (C.c_void_p * 15)(*[raw[f"sh{j + 1}"].ctypes.data if j < 12 else None for j in range(15)])
            L.gsplat_pack_sh_from_vec3(ptrs, n, sh[0].ctypes.data, sh[1].ctypes.data, sh[2].ctypes.data)
        else:
            ptrs = (C.c_void_p * 45)(*[raw[f"f_rest_{j}"].ctypes.data for j in range(45)])
            L.gsplat_pack_sh_from_frest(ptrs, n, sh[0].ctypes.data, sh[1].ctypes.data, sh[2].ctypes.data)
        host.shx, host.shy, host.shz = sh
    assert np.array_equal(q(edge)[:15], edge[:15].astype(np.float16).view(np.uint16))     # (the host quantiser itself, against numpy)
    cam = pkg.camera.make_camera(480, 320, sh_order=3 if scheme != "none" else 0, frame=5)
    a, b = pkg.Engine(0), pkg.Engine(0)
    try:
        a.upload(host, origin=(0.1, 0.2, -0.3))
        b.upload_raw(raw, origin=(0.1, 0.2, -0.3))
        ia, ib = a.render(cam), b.render(cam)
        assert np.array_equal(ia, ib, equal_nan=True) and ia[..., 3].max() > 0.5
        ra, rb = a.debug_records(n), b.debug_records(n)
        assert np.array_equal(ra.view(np.uint8), rb.view(np.uint8))
        # missing attributes: the reference's defaults on both paths
        bare_raw = {"P": s.P}
        bare = pkg.scenes.Splats(s.P, np.zeros((n, 3), np.uint16), np.ones(n, np.float32), np.full((n, 3), 0x3c00, np.uint16),
                                 np.tile(np.uint16([0, 0, 0, 0x3c00]), (n, 1)))
        cam0 = pkg.camera.make_camera(200, 150, sh_order=0, frame=1, distance=60.0)
        a.upload(bare); b.upload_raw(bare_raw)
        assert np.array_equal(a.render(cam0), b.render(cam0), equal_nan=True)
    finally:
        a.close(); b.close()


def test_rccl_entry_points_on_one_gpu(pkg, engine):
    """The 1-GPU box cannot host two RCCL ranks, but it can run the library's RCCL plumbing with world size 1: librccl is
    dlopen'ed, ncclGetUniqueId / ncclCommInitRank / ncclCommInitAll / ncclCommDestroy execute on the real GPU, and a frame
    through gsr_comm_render / gsr_multi_render equals the plain frame."""
    L = pkg.load_library()
    assert L.gsr_comm_available() == 1
    splats = pkg.scenes.make_scene(20000, seed=191, sh=True)
    cam = pkg.camera.make_camera(320, 200, sh_order=3, frame=2)
    engine.upload(splats)
    want = engine.render(cam)
    eng = pkg.Engine(0)
    hb = HipBuffers()
    try:
        eng.upload(splats)
        uid = eng.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        eng.comm_init(uid, 0, 1)
        out = hb.alloc(cam.height * cam.width * 16)
        eng.comm_render(pkg.engine.camera_struct(cam), out)
        eng.synchronize()
        assert np.array_equal(hb.download(out, (cam.height, cam.width, 4)), want)
        eng.comm_destroy()
    finally:
        hb.free()
        eng.close()
    with pkg.MultiEngine([0], pkg.engine.TRANSPORT_RCCL) as M:      # ncclCommInitAll over one device
        assert M.transport == pkg.engine.TRANSPORT_RCCL
        M.upload(splats)
        assert np.array_equal(M.render(cam), want)


# ---------------------------------------------------------------------------------------------
# round 4: the headline regime (culled, orbiting) verified at the size it is quoted on, and bench.py launching itself
@pytest.mark.parametrize("config", ["C4", "C5"])
def test_headline_regime_is_the_full_frame_at_full_size(pkg, oracle, config):
    """BASELINE C4 / C5 at FULL size: 12 frames of the bench's orbit plus one jump to the far side, the library's default policy
    (cluster culling, occlusion culling against the previous frame's horizons, small-frame sort, repairs) against a context
    that never culls anything: every frame bit for bit, most of them really culled -- and one culled frame against the oracle."""
    splats, cfg = pkg.scenes.make_config(config)
    w, h = cfg["width"], cfg["height"]
    frames = list(range(12)) + [60]                       # 3 degrees per frame; frame 60 = the other side of the cloud
    cams = [pkg.scenes.config_camera(config, pkg.camera, w, h, 3, i) for i in frames]
    plain, dflt = pkg.Engine(0), pkg.Engine(0)
    try:
        plain.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
        plain.set_option(pkg.engine.OPT_CLUSTER_CULL, 0)
        plain.upload(splats)
        dflt.upload(splats)
        checked = None
        for k, c in enumerate(cams):
            got = dflt.render(c)
            want = plain.render(c)
            assert np.array_equal(got, want), f"{config}: frame {frames[k]} differs from the frame rendered without culling"
            if k == 8:
                checked = (c, got.copy())
        st = dflt.stats()
        assert st["frames_culled"] >= 10, st                # (the first frames of a cloud are unculled: the policy needs their verdict)
        assert plain.stats()["frames_culled"] == 0
        # the jump: the policy sees it coming (the camera moved further than horizons tolerate) and does not even try the old
        # horizons -- a front-slab frame instead of a culled attempt plus a repair; pixels were compared above
        assert st["frames_jumped"] == 1 and st["frames_slab"] >= 1, st
        assert st["frames_repaired"] <= 1, st
        c, img = checked
        _check_image(img, oracle.render(splats, c, threads=oracle.max_threads()))
    finally:
        plain.close(); dflt.close()


def _run_bench(args, env_extra, timeout=900):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None)


def test_bench_launches_itself_on_several_gpus(pkg):
    """`python bench.py --gpus N` with no launcher drives gsr_multi_* from one process.  On this box the ranks share the GPU over
    the COPY transport (a functional run, flagged as such); the line is the driver's contract and the stitched frame is the
    unsharded frame bit for bit.  Without enough GPUs (and without the override) the line carries an error and the exit code says so."""
    ndev = int(pkg.load_library().gsr_device_count())
    res, line = _run_bench(["--gpus", "2", "--config", "C3", "--steps", "12", "--warmup", "4", "--no-cpu-baseline"],
                           {"GSR_BENCH_ALLOW_DUP": "1"})
    assert res.returncode == 0, res.stderr[-2000:]
    assert line is not None and line["n_gpus"] == 2 and line["sharded_frame_bit_identical"] is True and line["value"] > 0
    assert line["config"]["functional_only"] == (ndev < 2) and len(line["per_rank_stages_ms"]) == 2
    assert line["gather_ms"] is not None and line["gather_ms"] > 0 and len(line["rccl_ranks"]) == 2
    if ndev >= 2:
        assert line["rccl_ranks"] == [0, 1] and line["rccl_comm_count"] == 2
    # interleaved rows through the same door
    res, line = _run_bench(["--gpus", "3", "--config", "C2", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--shard-layout", "0", "--no-extra-legs"],
                           {"GSR_BENCH_ALLOW_DUP": "1"})
    assert res.returncode == 0 and line["sharded_frame_bit_identical"] is True, res.stderr[-2000:]
    if ndev < 64:
        res, line = _run_bench(["--gpus", "64", "--steps", "2", "--warmup", "1"], {"GSR_BENCH_ALLOW_DUP": "0"})
        assert res.returncode != 0 and line is not None and "error" in line and line["value"] is None


def test_bench_one_rank_through_the_multi_gpu_door_matches_the_single_context(pkg):
    """`bench.py --gpus 1 --via-multi` (gsr_multi with ONE rank: the N = 1 point of a scaling curve measured through `--gpus N`)
    must give the single-context line's frame rate -- so that, the day the curve is measured, its first point agrees with BENCH.
    Best of two runs each way (a box's runs differ by ~2 %); the gather line carries the per-link figures for N > 1."""
    argv = ["--config", "C3", "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--no-extra-legs", "--no-verify"]
    single, multi = [], []
    for _ in range(1):
        res, line = _run_bench(argv, {})
        assert res.returncode == 0 and line["n_gpus"] == 1, res.stderr[-2000:]
        single.append(line["value"])
        res, line = _run_bench(argv + ["--gpus", "1", "--via-multi"], {})
        assert res.returncode == 0 and line["n_gpus"] == 1 and "gsr_multi" in line["config"]["parallelism"], res.stderr[-2000:]
        multi.append(line["value"])
    # (a performance figure inside the correctness suite: wide enough not to flake on a shared or throttled GPU -- the two doors run the
    #  same kernels; a real regression of the multi-GPU door at N = 1 is tens of per cent, from an extra copy or a lost overlap)
    assert abs(max(multi) / max(single) - 1.0) < 0.15, (single, multi)
    res, line = _run_bench(["--gpus", "2", "--config", "C2", "--steps", "8", "--warmup", "3", "--no-cpu-baseline"], {"GSR_BENCH_ALLOW_DUP": "1"})
    assert res.returncode == 0, res.stderr[-2000:]
    gl = line["gather_links"]
    assert gl["bytes_per_peer"] == [line["config"]["width"] * 16 * sum(min(16, line["config"]["height"] - r * 16) for r in range(23, 45))]
    assert gl["GBps_per_link"] > 0 and gl["GBps_assumed_per_link"] == 153.0 and len(line["per_rank_ms_per_step"]) == 2


def test_bench_takes_an_inria_ply(pkg, oracle, tmp_path):
    """`bench.py --ply PATH`: an INRIA 3DGS capture file (what the reference's example scene imports) goes through ply.py's
    activations and gets the whole line -- frame rate, roofline, CPU baseline, the last timed frame checked bit for bit -- on an orbit
    fitted to the cloud; and that frame is the oracle's frame of the same splats and camera."""
    v = pkg.scenes.make_inria_raw(60000, seed=9, radius=0.8)
    v["x"] += 5.0; v["z"] -= 2.0                                   # (somewhere else than the origin: the orbit has to find it)
    path = str(tmp_path / "capture.ply")
    pkg.scenes.write_inria_ply(path, v)
    res, line = _run_bench(["--ply", path, "--steps", "30", "--warmup", "5", "--cpu-seconds", "4", "--no-other-configs"], {})
    assert res.returncode == 0, res.stderr[-2000:]
    assert line["value"] > 0 and line["timed_frame_bit_identical"] is True and line["data"].startswith("INRIA PLY")
    assert line["config"]["workload"].startswith("PLY capture.ply: 60000 splats") and line["config"]["n_splats"] == 60000
    assert line["n_visible"] > 5000 and line["cpu_baseline"]["value"] > 0 and line["roofline"]["pairs_consumed_per_launch"] > 0
    name = pkg.scenes.register_ply_config(path, pkg.ply, name="PLY_T")
    try:
        splats, cfg = pkg.scenes.make_config(name)
        cam = pkg.scenes.config_camera(name, pkg.camera, 480, 270, 3, 3)
        with pkg.Engine(0) as eng:
            eng.upload(splats)
            img = eng.render(cam)
        ref = oracle.render(splats, cam)
        assert np.abs(img - ref).max() <= 1e-3 and (ref[..., 3] > 0.5).mean() > 0.05
    finally:
        pkg.scenes.CONFIGS.pop("PLY_T", None)


def test_bench_single_gpu_line_says_whether_the_timed_frame_is_the_full_frame(pkg):
    res, line = _run_bench(["--config", "C3", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-other-configs"], {})
    assert res.returncode == 0, res.stderr[-2000:]
    assert line["timed_frame_bit_identical"] is True and line["occlusion_culling"]["frames_culled"] > 0
    assert line["roofline"]["peak_measured"] is None or line["roofline"]["peak_measured"] > 3000


def test_multi_gpu_gather_overlaps_the_next_frame_and_keeps_every_frame(pkg, engine):
    """gsr_multi with double-buffered bands: many device-target frames back to back into FEWER target buffers than frames (a
    target is reused while older gathers are still in flight), both layouts, resolution change in between -- every frame that is
    read back is the single-GPU frame"""
    splats = pkg.scenes.make_scene(200000, seed=313, sh=True)
    engine.upload(splats)
    hb = HipBuffers()
    try:
        with pkg.MultiEngine([0] * 3, pkg.engine.TRANSPORT_COPY) as M:
            M.upload(splats)
            for layout in (1, 0):
                M.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
                for (w, h) in ((640, 360), (500, 333)):
                    cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in range(9)]
                    want = [engine.render(c) for c in cams]
                    outs = [hb.alloc(w * h * 16) for _ in range(3)]
                    for k, c in enumerate(cams):
                        M.render_struct_to_device(pkg.engine.camera_struct(c), outs[k % 3], 0)
                        if k % 3 == 2:                                  # read the last three frames back
                            M.synchronize()
                            for j in range(3):
                                assert np.array_equal(hb.download(outs[j], (h, w, 4)), want[k - 2 + j]), (layout, w, k - 2 + j)
            ms, n = M.gather_stats(1)
            M.render(pkg.camera.make_camera(500, 333, sh_order=3, frame=3))
            ms, n = M.gather_stats(0)
            assert n == 1 and ms > 0
            assert M.comm_info() == ([-1, -1, -1], 0)
    finally:
        hb.free()


def test_small_frame_sort_backs_off_when_a_tie_run_defeats_it_every_frame(pkg):
    """A cloud with a run of more than 64 coincident splats: the small-frame sort's tie rule gives that bucket up, and the frame
    is rendered again with the global sort.  The re-render refills the hints, so without a back-off EVERY frame would be
    rendered twice; after three failures in a row the slot stays with the global sort for a while.  Pixels are the global
    sort's throughout."""
    splats = pkg.scenes.make_scene(60000, seed=401, sh=True)
    splats.P[5000:5100] = splats.P[4000]                       # 101 splats at one position: 101 equal sort keys
    w, h = 640, 400
    cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in range(16)]
    ref, eng = pkg.Engine(0), pkg.Engine(0)
    try:
        ref.set_option(pkg.engine.OPT_LOCAL_SORT, 0)
        ref.upload(splats); eng.upload(splats)
        for k, c in enumerate(cams):
            assert np.array_equal(eng.render(c), ref.render(c)), f"frame {k}"
        st = eng.stats()
        assert 1 <= st["frames_resorted"] <= 4, st["frames_resorted"]      # (16 without the back-off)
    finally:
        ref.close(); eng.close()


@pytest.mark.parametrize("shard", [(0, 1, 0), (1, 3, 1), (2, 4, 0)])
def test_front_slab_frames_are_exact(pkg, oracle, shard):
    """GSR_OPT_FRONT_SLAB: occlusion culling inside ONE frame.  The nearest splats (a slab picked from a histogram of the surviving
    clusters' distances) are composited first; the tiles that are opaque by then are finished, and the rest of the cloud is projected,
    sorted, binned and composited only where a tile is still open, continuing from the stored colour and transmittance.  No
    previous frame, no prediction, no check -- and every frame must equal the one-pass frame bit for bit: on an orbit, across
    jumps, from inside the cloud, depth-tested, sharded, with every slab size, and as the repair of a temporally culled frame."""
    idx, count, layout = shard
    splats = pkg.scenes.make_scene(400000, seed=197, sh=True, radius=1.0)
    w, h = 960, 540
    cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in (0, 1, 40, 41, 77)]
    cams += [pkg.camera.make_camera(w, h, sh_order=3, frame=43, distance=d) for d in (0.5, 2.2, 9.0)]          # inside, near, far
    cams += [pkg.camera.make_camera(w, h, sh_order=3, frame=9, distance=2.5, pivot=(1.5, 0.4, 0.0))]              # most of the cloud off screen
    cm = pkg.camera
    cams += [cm.make_camera(w, h, sh_order=3, frame=3, proj_matrix=cm.orthographic(-1.35, 1.35, -1.35 * h / w, 1.35 * h / w, 0.05, 60.0))]
    rng = np.random.default_rng(11)
    depth = np.where(rng.random((h, w)) < 0.5, 0.9973, 1.0).astype(np.float32)
    plain, eng = pkg.Engine(0), pkg.Engine(0)
    try:
        for e in (plain, eng):
            e.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
            e.set_row_shard(idx, count)
            e.upload(splats)
        plain.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
        want = [plain.render(c).copy() for c in cams]
        want_d = [plain.render_depth(c, depth).copy() for c in cams[:3]]
        vis_full = plain.stats()["n_visible"]
        if count == 1:
            _check_image(want[0], oracle.render(splats, cams[0], threads=oracle.max_threads()))
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 3)              # every frame a front-slab frame
        for k, (c, ref) in enumerate(zip(cams, want)):
            assert np.array_equal(eng.render(c), ref), f"front-slab frame {k} differs"
        for k, (c, ref) in enumerate(zip(cams[:3], want_d)):
            assert np.array_equal(eng.render_depth(c, depth), ref), f"depth-tested front-slab frame {k} differs"
        st = eng.stats()
        assert st["frames_slab"] == len(cams) + 3 and st["frames_culled"] == 0 and st["frames_repaired"] == 0, st
        assert np.array_equal(eng.render(cams[0]), want[0])
        assert eng.stats()["n_visible"] < 0.5 * vis_full              # (what phase 2 kept: far less than the frame holds)
    finally:
        eng.close()
    # every slab size, from "almost nothing in front" to "everything in front"
    import os
    for frac, mn in (("1", "1"), ("128", "64"), ("255", "100000000")):
        old = {k: os.environ.get(k) for k in ("GSR_SLAB_FRAC", "GSR_SLAB_MIN")}
        os.environ["GSR_SLAB_FRAC"], os.environ["GSR_SLAB_MIN"] = frac, mn
        try:
            e2 = pkg.Engine(0)
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        try:
            e2.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
            e2.set_row_shard(idx, count)
            e2.set_option(pkg.engine.OPT_OCCLUSION_CULL, 3)
            e2.upload(splats)
            for k in (0, 2, 5, 8):
                assert np.array_equal(e2.render(cams[k]), want[k]), f"slab {frac}/{mn}: frame {k} differs"
        finally:
            e2.close()
    # the default policy: temporal culling in the steady state, a front-slab frame where a horizon broke (the jumps)
    e3 = pkg.Engine(0)
    try:
        e3.set_option(pkg.engine.OPT_SHARD_LAYOUT, layout)
        e3.set_row_shard(idx, count)
        e3.set_option(pkg.engine.OPT_OCCLUSION_CULL, 2)
        e3.set_option(pkg.engine.OPT_FRONT_SLAB, 2)
        e3.upload(splats)
        orbit = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in (0, 1, 2, 3, 50, 51, 52, 110, 111)]
        for k, c in enumerate(orbit):
            assert np.array_equal(e3.render(c), plain.render(c)), f"default policy: frame {k} differs"
        st = e3.stats()
        assert st["frames_slab"] >= 1 and st["frames_culled"] >= 5, st
    finally:
        e3.close(); plain.close()


def test_a_rank_whose_rows_see_nothing(pkg):
    """a context that has never met a (super-tile, splat) pair -- one splat, a row shard that does not own its rows -- under forced
    occlusion culling and forced front-slab frames: the frame-end kernels read entry 0 of the list buffer unconditionally, and a slot
    without one faulted on the null pointer (found by tools/fuzz_parity.py).  And the padding of a band (rows behind the rank's
    last image row) reads as zeros in a host target."""
    E = pkg.engine
    splats = pkg.scenes.make_scene(1, seed=5, sh=False)
    cams = [pkg.camera.make_camera(64, 720, sh_order=0, frame=f, distance=4.61995 * d) for f, d in ((0, 1.0), (1, 1.0), (2, 1.0), (40, 1.3))]
    for cull, slab in ((1, 2), (2, 1), (2, 2), (3, 1)):
        eng = E.Engine(0)
        try:
            eng.set_option(E.OPT_SHARD_LAYOUT, 0)
            eng.set_row_shard(2, 3)
            eng.set_option(E.OPT_OCCLUSION_CULL, cull)
            eng.set_option(E.OPT_FRONT_SLAB, slab)
            eng.set_option(E.OPT_LOCAL_SORT, 2)
            eng.set_option(E.OPT_FRAMES_IN_FLIGHT, 2)
            eng.upload(splats)
            for c in cams:
                assert not eng.render(c).any()
        finally:
            eng.close()
    # more ranks than tile rows (48 pixels = 3 tile rows, 8 contiguous bands): the last ranks own nothing; every band is defined
    big = pkg.scenes.make_scene(20000, seed=6, sh=False)
    cam = pkg.camera.make_camera(1920, 48, sh_order=0, frame=0)
    full = E.Engine(0)
    try:
        full.upload(big)
        img = full.render(cam)
        assert img[..., 3].max() > 0.5
        for idx in (0, 2, 3, 7):
            eng = E.Engine(0)
            try:
                eng.set_option(E.OPT_SHARD_LAYOUT, 1)
                eng.set_row_shard(idx, 8)
                eng.upload(big)
                for _ in range(2):
                    assert np.array_equal(eng.render(cam), pkg.multigpu.extract_band(img, idx, 8, 1)), idx
            finally:
                eng.close()
    finally:
        full.close()


def test_front_slab_phase_two_outgrows_the_list_buffer(pkg):
    """phase 2 of a front-slab frame continues from what phase 1 composited: when its lists outgrow the buffer (sized by frames
    that showed far less), the clamped speculative back end must not simply run again on top of its own output -- the whole
    frame is rendered again"""
    E = pkg.engine
    splats = pkg.scenes.make_scene(20000, seed=77, sh=True)
    # (large splats: from afar a splat reaches one or two super-tiles, from near by most of the 135 -- far more pairs than the two
    #  entries per splat a slot's first buffer allows for)
    splats.scale[:] = pkg.scenes.f16bits(np.random.default_rng(7).uniform(0.05, 0.5, size=(splats.n, 3)))
    far = [pkg.camera.make_camera(960, 540, sh_order=3, frame=f, distance=200.0) for f in (0, 1)]
    near = [pkg.camera.make_camera(960, 540, sh_order=3, frame=f) for f in (30, 31)]
    dut, plain = E.Engine(0), E.Engine(0)
    try:
        plain.set_option(E.OPT_OCCLUSION_CULL, 0)
        dut.set_option(E.OPT_OCCLUSION_CULL, 3)
        dut.set_option(E.OPT_FRONT_SLAB, 2)
        plain.upload(splats); dut.upload(splats)
        for c in far + near + far + near:
            assert np.array_equal(dut.render(c), plain.render(c))
        st = dut.stats()
        assert st["frames_slab"] >= 6 and st["frames_requeued"] >= 1, st
    finally:
        dut.close(); plain.close()


def test_randomised_frames_match_the_oracle(pkg, oracle):
    """random small clouds (scale ranges, huge splats, coincident splats), framebuffers and cameras -- perspective, off-centre,
    orthographic, object-level transforms, near / far planes that cut the cloud, cameras inside it -- under random library
    options: every frame within 1e-3 per channel of the CPU oracle (the generators are tools/fuzz_parity.py's)"""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(root, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    E = pkg.engine
    rng = np.random.default_rng(int(os.environ.get("GSR_FUZZ_SEED", "2026")))
    worst = 0.0
    for it in range(int(os.environ.get("GSR_FUZZ_ITERS", "40"))):      # (a longer hunt: GSR_FUZZ_ITERS=400 GSR_FUZZ_SEED=...)
        n = int(rng.choice([1, 64, 65, 1000, 5000, 20000, 60000]))
        sh = bool(rng.integers(0, 2))
        lo = rng.uniform(-6.0, -3.0)
        splats = pkg.scenes.make_scene(n, seed=int(rng.integers(1, 1 << 30)), sh=sh, log_scale_range=(lo, lo + rng.uniform(0.5, 2.5)))
        if n >= 1000 and rng.random() < 0.3:
            splats.scale[:100] = pkg.scenes.f16bits(rng.uniform(0.2, 1.5, size=(100, 3)))
        if n >= 1000 and rng.random() < 0.3:
            splats.P[100:140] = splats.P[100]
        w, h = int(rng.choice([64, 333, 640, 1280])), int(rng.choice([48, 217, 480]))
        order = int(rng.integers(0, 4)) if sh else 0
        kind = int(rng.integers(0, 3))
        cams = [fz.random_camera(np.random.default_rng(5000 + it), w, h, order, f, d, kind) for f, d in ((0, 1.0), (1, 1.0), (40, 1.3))]
        eng = E.Engine(0)
        try:
            eng.set_option(E.OPT_OCCLUSION_CULL, int(rng.choice([0, 1, 2, 3])))
            eng.set_option(E.OPT_FRONT_SLAB, int(rng.choice([0, 1, 2])))
            eng.set_option(E.OPT_LOCAL_SORT, int(rng.choice([0, 1, 2])))
            eng.set_option(E.OPT_LAZY_COLOUR, int(rng.choice([0, 1, 2])))
            eng.upload(splats)
            for k, c in enumerate(cams):
                img = eng.render(c)
                if k != 1:
                    ref = oracle.render(splats, c, threads=oracle.max_threads())
                    worst = max(worst, _check_image(img, ref))
        finally:
            eng.close()
    assert worst > 0.0     # (something was drawn)


def test_randomised_wire_overlays_match_the_oracle(pkg, oracle):
    """the wireframe overlay (N3) of random clouds under random cameras: bit-identical to the oracle's"""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(root, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    rng = np.random.default_rng(int(os.environ.get("GSR_FUZZ_SEED", "77")))
    eng = pkg.Engine(0)
    try:
        drawn = 0
        for it in range(int(os.environ.get("GSR_FUZZ_ITERS", "25"))):
            n = int(rng.choice([1, 65, 1000, 20000]))
            lo = rng.uniform(-6.0, -2.5)
            splats = pkg.scenes.make_scene(n, seed=int(rng.integers(1, 1 << 30)), sh=bool(rng.integers(0, 2)), log_scale_range=(lo, lo + rng.uniform(0.5, 2.5)))
            w, h = int(rng.choice([64, 333, 640])), int(rng.choice([48, 217, 400]))
            cam = fz.random_camera(np.random.default_rng(9000 + it), w, h, 0, int(rng.integers(0, 60)), float(rng.choice([1.0, 1.3, 0.6])), int(rng.integers(0, 3)))
            eng.upload(splats)
            img = eng.render_wire(cam)
            assert np.array_equal(img, oracle.render_wire(splats, cam)), (it, n, w, h)
            drawn += int(img.any())
        assert drawn >= 8
    finally:
        eng.close()


def test_randomised_records_and_depth_order_match_the_oracle(pkg, oracle):
    """K1's records and sort keys bit for bit, and the depth order index for index, against the oracle under random cameras
    (perspective / off-centre / orthographic, object-level transforms, near / far planes, positions inside the cloud), random
    origins and clouds with coincident splats"""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(root, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    rng = np.random.default_rng(int(os.environ.get("GSR_FUZZ_SEED", "4242")))
    eng = pkg.Engine(0)
    try:
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)      # (every clip-visible splat gets a record)
        seen = 0
        for it in range(int(os.environ.get("GSR_FUZZ_ITERS", "30"))):
            n = int(rng.choice([65, 1000, 20000, 70001]))
            sh = bool(rng.integers(0, 2))
            lo = rng.uniform(-6.0, -3.0)
            splats = pkg.scenes.make_scene(n, seed=int(rng.integers(1, 1 << 30)), sh=sh, log_scale_range=(lo, lo + rng.uniform(0.5, 2.5)))
            if n >= 1000:
                splats.P[100:160] = splats.P[500:560]          # exact ties
            w, h = int(rng.choice([64, 333, 640])), int(rng.choice([48, 217, 400]))
            order = int(rng.integers(0, 4)) if sh else 0
            cam = fz.random_camera(np.random.default_rng(7000 + it), w, h, order, int(rng.integers(0, 60)), float(rng.choice([1.0, 1.3, 0.5])), int(rng.integers(0, 3)))
            origin = tuple(float(v) for v in rng.uniform(-0.5, 0.5, 3)) if rng.random() < 0.5 else (0.0, 0.0, 0.0)
            eng.upload(splats, origin=origin)
            eng.render(cam)
            dev = eng.debug_records(splats.n)
            ref = oracle.preprocess(splats, cam, origin=origin)
            vis = dev["visible"] == 1
            seen += int(vis.sum())
            assert (ref["visible"][vis] == 1).all(), it
            assert np.array_equal(dev["key"][vis].view(np.uint32), ref["key"][vis].view(np.uint32)), it
            for f in REC_FIELDS:
                a, b = dev[f][vis].view(np.uint32), ref[f][vis].view(np.uint32)
                assert np.array_equal(a, b), f"iteration {it}, field {f}: {np.count_nonzero(a != b)} mismatches"
            order_dev = eng.debug_depth_order(splats.n)
            order_ref = oracle.host_sort_only(splats.P, cam.cam_pos)
            keep = np.zeros(splats.n, bool)
            keep[order_dev] = True
            assert np.array_equal(order_dev, order_ref[keep[order_ref]]), it
        assert seen > 1000
    finally:
        eng.close()


def test_randomised_shim_protocol(pkg):
    """the host shim (N2) driven like Houdini drives the reference -- primitives registered, re-cooked (new cache version),
    destroyed, shown in changing subsets -- against the C ABI fed by hand: every redraw = the active entries in registry (id)
    order, SH present iff the LAST active entry has it (entries without get zeros), the shim's own origin"""
    import os
    E = pkg.engine
    rng = np.random.default_rng(int(os.environ.get("GSR_FUZZ_SEED", "31")))
    R = pkg.GSplatRenderer(0)
    direct = E.Engine(0)
    S = pkg.scenes.Splats
    try:
        R.setSphericalHarmonicsOrder(3)
        details = {}          # gdp -> (id, splats)
        version = 1
        drawn = 0
        for step in range(int(os.environ.get("GSR_FUZZ_ITERS", "60"))):
            act = rng.random()
            gdp = 0x1000 * int(rng.integers(1, 6))
            if act < 0.35 or not details:                       # a primitive cooks (new or re-cooked: a new cache version purges the old entry)
                version += 1
                sp = pkg.scenes.make_scene(int(rng.choice([1, 64, 500, 3000, 20000])), seed=int(rng.integers(1, 1 << 30)), sh=bool(rng.random() < 0.6))
                sp.P[:] += rng.uniform(-0.6, 0.6, 3).astype(np.float32)
                details[gdp] = (R.registerUpdate(gdp, (version, 0, 0, 0), 0, sp), sp)
            elif act < 0.45 and gdp in details:                 # a primitive is destroyed
                R.flushEntriesForMatchingDetail(details.pop(gdp)[0])
            w, h = int(rng.choice([64, 320, 640])), int(rng.choice([48, 240, 400]))
            cam = pkg.camera.make_camera(w, h, sh_order=3, frame=int(rng.integers(0, 90)), distance=4.61995 * float(rng.choice([1.0, 1.3, 0.7])))
            ids = sorted(rid for rid, _ in details.values())
            active = [rid for rid in ids if rng.random() < 0.7]
            img = R.frame(cam, active)
            if not active:
                assert not img.any(), step
                continue
            by_id = {rid: sp for rid, sp in details.values()}
            parts = [by_id[rid] for rid in active]
            has_sh = parts[-1].shx is not None                   # registry iteration order = id order: the last active entry decides
            n_all = sum(p_.n for p_ in parts)
            def field(f):
                return np.concatenate([getattr(p_, f) for p_ in parts])
            def shf(f):
                return np.concatenate([getattr(p_, f) if getattr(p_, f) is not None else np.zeros((p_.n, 16), np.uint16) for p_ in parts]) if has_sh else None
            cat = S(P=field("P"), Cd=field("Cd"), alpha=field("alpha"), scale=field("scale"), orient=field("orient"), shx=shf("shx"), shy=shf("shy"), shz=shf("shz"))
            assert R.query(R.Q_SPLAT_COUNT) == n_all and R.query(R.Q_SH_PRESENT) == int(has_sh), step
            direct.upload(cat, origin=tuple(float(v) for v in R.origin()))
            cam_d = pkg.camera.make_camera(w, h, sh_order=3 if has_sh else 0, frame=cam.meta["frame"], distance=cam.meta["distance"])
            cam_d.cam_pos = R.lastCameraPos()            # (the shim's own float32 inverse of the view matrix: src/GSplatRenderer.C:551-563)
            ref = direct.render(cam_d)
            err = float(np.abs(img - ref).max())
            assert err == 0.0, (step, err, active, np.abs(cam_d.cam_pos - cam.cam_pos).max())
            drawn += int(img.any())
        assert drawn >= 10
    finally:
        R.close(); direct.close()


def test_randomised_exactness_soak(pkg):
    """tools/fuzz_parity.py, a short run: random clouds, framebuffers, projections, row shards and library options; every frame of a
    short camera path bit-identical to a context that culls nothing, takes the global sort and shades eagerly"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # fresh contexts per iteration; then ONE long-lived context with re-uploads and option / shard flips in mid-stream
    for args in (["120", "11"], ["100", "12", "0", "2"]):
        res = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py")] + args, capture_output=True, text=True, timeout=900, cwd=root)
        assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
        assert "frames bit-identical" in res.stdout


def test_a_frame_whose_clusters_are_all_culled_is_empty(pkg, engine):
    """every cluster off screen / behind the eye: no K1 slot is filled and no sort workgroup runs -- the frame must be EMPTY, not the
    previous frame's splats walked again (the sorted count of a slot outlives its frame)"""
    splats = pkg.scenes.make_scene(200000, seed=5, sh=True)
    cam = pkg.camera.make_camera(640, 360, sh_order=3, frame=2)
    away = pkg.camera.make_camera(640, 360, sh_order=3, frame=2, pivot=(0.0, 0.0, 0.0), distance=-30.0)     # the cloud behind the eye
    engine.upload(splats)
    for mode in (0, 1, 2, 3):
        engine.set_option(pkg.engine.OPT_OCCLUSION_CULL, mode)
        try:
            a = engine.render(cam)
            assert a[..., 3].max() > 0.5
            b = engine.render(away)
            assert not b.any(), f"occlusion culling mode {mode}: {np.count_nonzero(b)} non-zero values in a frame that shows nothing"
            assert engine.stats()["n_visible"] == 0
            assert np.array_equal(engine.render(cam), a)
        finally:
            engine.set_option(pkg.engine.OPT_OCCLUSION_CULL, 1)


def _zwin_quantiles(cam, P, qs):
    """window depths (0..1) of the splat centres the camera can see: where an opaque pass's depth buffer has to sit to cut the cloud"""
    M = np.asarray(cam.proj, np.float64).reshape(4, 4).T @ np.asarray(cam.obj_view, np.float64).reshape(4, 4).T
    Ph = np.concatenate([P[:20000].astype(np.float64), np.ones((min(len(P), 20000), 1))], axis=1) @ M.T
    ok = (Ph[:, 3] > 0) & (np.abs(Ph[:, 2]) <= Ph[:, 3])
    zw = 0.5 * Ph[ok, 2] / Ph[ok, 3] + 0.5
    return [float(np.quantile(zw, q)) for q in qs]


@pytest.mark.gpu
def test_depth_culling_and_quadrant_classification_are_invisible(pkg, oracle):
    """Round 6: a depth-tested frame (the one the viewport hook issues on every redraw) is (a) classified per 8x8 quadrant inside k_blend --
    a quadrant under a buffer cleared to the far plane runs the plain loop, a record behind everything under it is not staged, one in
    front of everything needs no per-pixel compare -- and (b) culled in K1 / k_cluster_cull against the tile-max pyramid of the depth
    buffer.  Neither may change a pixel: every frame is compared bit for bit with a context that does none of it (GSR_FLAG_NO_DEPTH_CLASS,
    GSR_OPT_OCCLUSION_CULL = 0, GSR_OPT_CLUSTER_CULL = 0: every record staged, every fragment compared), over depth buffers of every
    kind -- far plane, a sphere among the splats, tile-sized and odd-sized blocks, per-pixel noise, NaN, negative, beyond 1 -- at
    odd framebuffer sizes, sharded, as front-slab frames, and with the camera moving under a changing depth buffer."""
    E = pkg.engine
    rng = np.random.default_rng(606)
    for (n, w, h) in ((300000, 1280, 720), (120000, 1001, 517), (60000, 330, 250)):
        splats = pkg.scenes.make_scene(n, seed=77 + n % 13, sh=True)
        cams = [pkg.camera.make_camera(w, h, sh_order=2, frame=i) for i in (0, 1, 2, 3, 30, 31)]
        q = _zwin_quantiles(cams[0], splats.P, (0.05, 0.2, 0.4, 0.6, 0.8, 0.95))
        yy, xx = np.mgrid[0:h, 0:w]
        depths = {"far": np.ones((h, w), np.float32),
                  "sphere": pkg.scenes.sphere_occluder_depth(cams[0], 3.42, 0.645),
                  "blocks16": np.asarray(q + [1.0, 1.0], np.float32)[rng.integers(0, 8, ((h + 15) // 16, (w + 15) // 16))].repeat(16, 0).repeat(16, 1)[:h, :w].copy(),
                  "blocks40": np.asarray(q + [1.0, 0.0, 1.5, -0.5, np.nan], np.float32)[rng.integers(0, 11, ((h + 39) // 40, (w + 39) // 40))].repeat(40, 0).repeat(40, 1)[:h, :w].copy(),
                  "noise": np.where(rng.random((h, w)) < 0.5, np.float32(q[2]), np.float32(1.0)).astype(np.float32),
                  "ramp": (q[0] + (q[5] - q[0]) * (xx / max(w - 1, 1))).astype(np.float32)}
        plain, dut = pkg.Engine(0), pkg.Engine(0)
        try:
            plain.set_option(E.OPT_OCCLUSION_CULL, 0); plain.set_option(E.OPT_CLUSTER_CULL, 0); plain.set_option(E.OPT_DEBUG_FLAGS, 32)
            plain.upload(splats); dut.upload(splats)
            for mode in ((1, 1), (2, 1), (3, 2), (0, 0)):          # (occlusion culling, front slab)
                dut.set_option(E.OPT_OCCLUSION_CULL, mode[0]); dut.set_option(E.OPT_FRONT_SLAB, mode[1])
                for name, d in depths.items():
                    for k, c in enumerate(cams):
                        got, want = dut.render_depth(c, d), plain.render_depth(c, d)
                        assert np.array_equal(got, want, equal_nan=True), f"{n} {w}x{h} mode {mode} depth {name} frame {k}: differs from the unclassified, unculled frame"
                # the depth buffer CHANGES under a moving camera (the opaque geometry is animated): no frame may use the previous buffer
                for k, c in enumerate(cams[:4]):
                    d = depths[("sphere", "far", "blocks16", "ramp")[k]]
                    assert np.array_equal(dut.render_depth(c, d), plain.render_depth(c, d), equal_nan=True), f"changing depth buffer, frame {k}"
                # ... nor under a STILL camera: a static redraw reuses the depth order only of a frame that was not culled against its buffer
                for name in ("sphere", "far", "ramp", "far", "blocks16"):
                    assert np.array_equal(dut.render_depth(cams[2], depths[name]), plain.render_depth(cams[2], depths[name]), equal_nan=True), f"still camera, depth {name}"
            # sharded (the depth buffer is the full image)
            dut.set_option(E.OPT_OCCLUSION_CULL, 1); dut.set_option(E.OPT_FRONT_SLAB, 1)
            for layout in (0, 1):
                for e in (dut, plain):
                    e.set_option(E.OPT_SHARD_LAYOUT, layout); e.set_row_shard(1, 3)
                for name in ("sphere", "blocks40"):
                    for c in cams[:3]:
                        assert np.array_equal(dut.render_depth(c, depths[name]), plain.render_depth(c, depths[name]), equal_nan=True), f"sharded {layout} {name}"
            for e in (dut, plain):
                e.set_row_shard(0, 1)
            # and the culling really happens: behind the sphere far fewer splats reach the sort than under the far plane
            for c in cams[:3]:
                dut.render_depth(c, depths["far"])
            v_far = dut.stats()["n_visible"]
            for c in cams[:3]:
                dut.render_depth(c, depths["sphere"])
            st = dut.stats()
            assert st["policy_bits"] & 32, st
            if n >= 300000:
                ramp0 = np.full((h, w), q[0], np.float32)       # an opaque wall in front of 95 % of the cloud
                for c in cams[:3]:
                    dut.render_depth(c, ramp0)
                assert dut.stats()["n_visible"] * 4 < v_far, (dut.stats()["n_visible"], v_far)
            # against the oracle
            if n == 120000:
                _check_image(dut.render_depth(cams[1], depths["sphere"]), oracle.render_depth(splats, cams[1], depths["sphere"]))
                _check_image(dut.render_depth(cams[1], depths["blocks16"]), oracle.render_depth(splats, cams[1], depths["blocks16"]))
        finally:
            plain.close(); dut.close()


@pytest.mark.gpu
def test_depth_tested_headline_frame_at_full_size(pkg, oracle):
    """BASELINE C4 at FULL size, depth-tested the way bench.py's `depth_tested` leg does it (a buffer cleared to the far plane; an opaque
    sphere among the splats covering 30 % of the frame): an orbit under the library's default policy against a context that culls
    and classifies nothing, bit for bit; the far-plane frames equal to the plain gsr_render frames; one frame of each against the oracle."""
    E = pkg.engine
    splats, cfg = pkg.scenes.make_config("C4")
    w, h = cfg["width"], cfg["height"]
    cams = [pkg.scenes.config_camera("C4", pkg.camera, w, h, 3, i) for i in range(8)]
    far = np.ones((h, w), np.float32)
    occ = pkg.scenes.sphere_occluder_depth(cams[0], 3.42, 0.645)       # wholly under the cloud's surface (0.16 units at its nearest)
    vis = pkg.scenes.sphere_occluder_depth(cams[0], 3.0, 0.566)         # bench.py's: its front pokes out of the cloud, its rim lies under the surface
    plain, dflt = pkg.Engine(0), pkg.Engine(0)
    try:
        plain.set_option(E.OPT_OCCLUSION_CULL, 0); plain.set_option(E.OPT_CLUSTER_CULL, 0); plain.set_option(E.OPT_DEBUG_FLAGS, 32)
        plain.upload(splats); dflt.upload(splats)
        keep = {}
        for name, d in (("far", far), ("occluder", occ), ("visible", vis)):
            for k, c in enumerate(cams):
                got, want = dflt.render_depth(c, d), plain.render_depth(c, d)
                assert np.array_equal(got, want), f"C4 depth-tested ({name}) frame {k} differs from the unclassified, unculled frame"
                if name == "far":
                    assert np.array_equal(got, dflt.render(c)) if k == 3 else True
                if k == 6:
                    keep[name] = got.copy()
        st = dflt.stats()
        assert st["frames_culled"] >= 8, st
        _check_image(keep["visible"], oracle.render_depth(splats, cams[6], vis))
        assert np.abs(keep["visible"] - keep["far"]).max() > 0.05           # (this one shows)
        # (the sphere sits 0.16 units under the cloud's surface: every pixel over it saturates in front of it -- the frame is the far-plane
        #  frame to within the kernel's 2^-14 early-out -- but nothing in the library may ASSUME that: the tiles over it end their lists there)
        assert np.abs(keep["occluder"] - keep["far"]).max() <= 2.0 ** -13
    finally:
        plain.close(); dflt.close()


def test_upload_reports_its_stages_and_restaging_is_exact(pkg, oracle):
    """Round 6: an upload is host-to-device copies into one arena, then ONE ordering + ONE packing kernel at gsr_upload_end; gsr_stats says
    what each stage took.  Re-staging -- the same context, clouds of other sizes, with and without SH, in several entries, and back -- leaves
    exactly the cloud a fresh context holds (the arena and the sort scratch are kept between uploads)."""
    E = pkg.engine
    cam = pkg.camera.make_camera(640, 360, sh_order=3, frame=1)
    eng = E.Engine(0)
    try:
        for k, (n, sh) in enumerate(((120000, True), (30000, False), (250000, True), (64, True), (250001, False), (120000, True))):
            s = pkg.scenes.make_scene(n, seed=300 + k, sh=sh)
            if k % 2:
                cut = n // 3
                S = pkg.scenes.Splats
                parts = [S(*[None if getattr(s, f) is None else getattr(s, f)[a:b] for f in ("P", "Cd", "alpha", "scale", "orient", "shx", "shy", "shz")]) for a, b in ((0, cut), (cut, n))]
                eng.upload_parts(parts)
            else:
                eng.upload(s)
            st = eng.stats()
            assert st["uploads"] == k + 1 and st["n_splats"] == n
            um = st["upload_ms"]
            assert um[3] > 0 and um[0] > 0 and um[2] > 0 and um[0] + um[1] + um[2] <= um[3] * 1.05 + 0.5, um
            fresh = E.Engine(0)
            try:
                fresh.upload(s)
                assert np.array_equal(eng.render(cam), fresh.render(cam)), f"upload {k}: the re-staged cloud is not the freshly staged one"
                assert np.array_equal(eng.debug_storage_order(n), fresh.debug_storage_order(n))
            finally:
                fresh.close()
        _check_image(eng.render(cam), oracle.render(s, cam))
    finally:
        eng.close()


def test_bench_line_carries_the_depth_tested_and_boundary_legs(pkg):
    """the default bench line (a small config here): `depth_tested` -- plain / a cleared depth buffer / an opaque sphere, each checked bit for
    bit against a context that culls and classifies nothing -- and `boundary` -- the redraw through the nine verbs, and the re-stage loop"""
    res, line = _run_bench(["--config", "C2", "--steps", "24", "--warmup", "6", "--no-cpu-baseline", "--no-other-configs"], {})
    assert res.returncode == 0, res.stderr[-2000:]
    dt = line["depth_tested"]
    for k in ("plain", "far_plane", "occluder"):
        assert dt[k]["value"] > 0 and dt[k]["last_frame_bit_identical_to_unculled"] is True, (k, dt[k])
    assert dt["occluder"]["opaque_pixels_frac"] > 0.1 and dt["occluder"]["depth_culling_active"] is True and dt["far_plane"]["depth_culling_active"] is False
    b = line["boundary"]
    assert b["via_shim"]["value"] > 0 and b["via_shim"]["frame_within_1e-3_of_unculled"] is True and b["via_shim"]["stagings"] == 1
    r = b["restage"]
    assert r["value"] > 0 and r["stagings"] >= r["steps"] and r["upload_ms"]["host_to_device"] > 0 and r["upload_ms"]["device_side"] > 0


@pytest.mark.gpu
def test_scan_time_depth_filter_is_invisible_and_really_filters(pkg):
    """Round 6: in depth-tested frames of clouds below 2^23 splats the nine spare bits of a list entry's splat index carry a coarse window
    depth (gsr_zq), and a tile drops, while it scans, the entries behind everything the opaque pass left under its live pixels.  The
    codes are a monotone lower bound, so nothing that could contribute is dropped: frames are bit-identical with the codes off
    (GSR_FLAG_NO_ZCODES = 64) -- and the blend kernel gathers far fewer records under an occluder.  The list read-back masks the bits."""
    E = pkg.engine
    splats = pkg.scenes.make_scene(400000, seed=91, sh=True)
    w, h = 1280, 720
    cams = [pkg.camera.make_camera(w, h, sh_order=2, frame=i) for i in range(8)]
    q = _zwin_quantiles(cams[0], splats.P, (0.1, 0.3))
    depths = {"sphere": pkg.scenes.sphere_occluder_depth(cams[0], 3.42, 0.645),
              "wall": np.full((h, w), q[1], np.float32),
              "half": np.where(np.arange(w)[None, :] < w // 2, np.float32(q[0]), np.float32(1.0)).repeat(h, 0).astype(np.float32)}
    on, off = pkg.Engine(0), pkg.Engine(0)
    try:
        off.set_option(E.OPT_DEBUG_FLAGS, 64)
        on.upload(splats); off.upload(splats)
        for name, d in depths.items():
            g_on = g_off = 0
            for k, c in enumerate(cams):
                a, b = on.render_depth(c, d), off.render_depth(c, d)
                assert np.array_equal(a, b, equal_nan=True), f"{name} frame {k}: the depth codes changed a pixel"
                if k >= 3:       # (the first frames of a buffer run without codes: the library has not seen geometry in it yet)
                    g_on += on.stats()["pairs_consumed"]; g_off += off.stats()["pairs_consumed"]
            assert on.stats()["n_visible"] == off.stats()["n_visible"]
            # (under a flat wall K1's own test against the tile-max depth is exact: nothing is left to filter.  Under a curved surface it is not)
            assert g_on <= g_off and (name != "sphere" or g_on < 0.8 * g_off), (name, g_on, g_off)
        # the debug read-back of the lists hands out splat indices, not index words
        on.render_depth(cams[0], depths["sphere"]); off.render_depth(cams[0], depths["sphere"])
        la = on.debug_tile_lists()
        assert la[2].size > 0 and 0 <= int(la[2].min()) and int(la[2].max()) < splats.P.shape[0]
    finally:
        on.close(); off.close()


@pytest.mark.gpu
def test_host_target_frames_in_bands_are_the_same_frames(pkg, monkeypatch):
    """Round 6: a frame rendered into a HOST buffer is composited in bands of tile rows (the blend launch once per band, a workgroup of another
    band's tile leaving at once) and each band's rows are copied back on a second stream while the next ones composite.  Same pixels as
    the one-launch, one-copy form (GSR_HOST_BANDS=1), in every regime: temporal culling, front-slab frames, one-pass frames with lazy colour
    (the fallback kernel runs per band too), depth-tested frames (the guarded plain kernel and its re-queue), odd framebuffer heights."""
    E = pkg.engine
    splats = pkg.scenes.make_scene(250000, seed=5, sh=True)
    monkeypatch.setenv("GSR_HOST_BANDS", "1")
    one = pkg.Engine(0)
    monkeypatch.setenv("GSR_HOST_BANDS", "4")
    four = pkg.Engine(0)
    monkeypatch.setenv("GSR_HOST_BANDS", "7")
    seven = pkg.Engine(0)
    try:
        for e in (one, four, seven):
            e.upload(splats)
        for (w, h) in ((1280, 720), (1001, 517), (640, 1100)):
            cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in (0, 1, 2, 3, 40, 41)]
            sphere = pkg.scenes.sphere_occluder_depth(cams[0], 3.42, 0.645)
            far = np.ones((h, w), np.float32)
            for mode in ((1, 1), (3, 2), (0, 0)):
                for e in (one, four, seven):
                    e.set_option(E.OPT_OCCLUSION_CULL, mode[0]); e.set_option(E.OPT_FRONT_SLAB, mode[1])
                for k, c in enumerate(cams):
                    ref = one.render(c)
                    assert np.array_equal(ref, four.render(c)) and np.array_equal(ref, seven.render(c)), f"{w}x{h} mode {mode} frame {k}"
                for k, (c, d) in enumerate(zip(cams, (far, far, sphere, sphere, far, sphere))):
                    ref = one.render_depth(c, d)
                    assert np.array_equal(ref, four.render_depth(c, d)) and np.array_equal(ref, seven.render_depth(c, d)), f"{w}x{h} mode {mode} depth frame {k}"
    finally:
        one.close(); four.close(); seven.close()

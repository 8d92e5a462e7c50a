"""Randomised exactness soak (GPU box): python tools/fuzz_parity.py [iterations] [seed] [first] [mode]
Every iteration: a random cloud (0 .. 400 k splats, random scale range, SH or not, now and then huge, coincident or poisoned --
NaN / infinite / zero -- attributes), a random framebuffer (up to 16384 wide), a random projection (perspective / off-centre /
orthographic; object-level transforms, near / far planes that cut the cloud, cameras inside it), a random row shard and random
library options (occlusion culling mode, front slab, small-frame sort, lazy colour, frames in flight, cluster culling, storage
order, tile order, dilation, sort cache, super-tile edge, deferred check), then a short camera path (small steps, a jump, a
repeat, turns in place) -- every frame must be BIT-IDENTICAL to the same camera from a context that culls nothing, takes the
global sort and shades eagerly (that configuration is what the -m gpu tests hold against the CPU oracle).
Some iterations render depth-tested (an opaque pass's depth image in front of part of the frame), some stage the cloud as several
entries, some go through gsr_multi_* (several contexts on this GPU, COPY transport: shard, render, gather).
mode 1: clouds of 1 - 2.5 M splats at 1920x1080 (the policy's temporal culling and front-slab frames engage by themselves).
mode 2: ONE long-lived context under test for the whole run: re-uploads, option flips and shard changes in mid-stream.
mode 3: the same with a long-lived gsr_multi of three contexts.
At the first difference the iteration is replayed on fresh contexts -- alone, then behind its predecessors, then with every
non-default option put back -- to say what it takes, and the run exits non-zero with the configuration that produced it."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
E = pkg.engine
HIP = C.CDLL("libamdhip64.so")

DEFAULTS = {E.OPT_OCCLUSION_CULL: 1, E.OPT_FRONT_SLAB: 1, E.OPT_LOCAL_SORT: 1, E.OPT_LAZY_COLOUR: 1, E.OPT_FRAMES_IN_FLIGHT: 1, E.OPT_CLUSTER_CULL: 1,
            E.OPT_STORAGE_ORDER: 1, E.OPT_XCD_SWIZZLE: 2, E.OPT_CULL_DILATE: 2, E.OPT_SORT_CACHE: 1, E.OPT_SUPER_TILE: 0, E.OPT_DEFERRED_CHECK: 0}
FLIPS = [(E.OPT_OCCLUSION_CULL, [0, 1, 2, 3]), (E.OPT_FRONT_SLAB, [0, 1, 2]), (E.OPT_LOCAL_SORT, [0, 1, 2]), (E.OPT_LAZY_COLOUR, [0, 1, 2]),
         (E.OPT_FRAMES_IN_FLIGHT, [1, 2]), (E.OPT_CLUSTER_CULL, [0, 1]), (E.OPT_XCD_SWIZZLE, [0, 1, 2, 3]), (E.OPT_CULL_DILATE, [0, 1, 2, 5])]


def random_camera(rng, w, h, order, frame, dist_scale, kind):
    # (rng is seeded per iteration: the draws below are the same for every frame of the iteration's path)
    near, far = float(rng.choice([0.01, 0.01, 0.5, 2.0])), float(rng.choice([1.0e5, 1.0e5, 6.0, 4.7]))
    obj = None
    if rng.random() < 0.3:                      # an object-level transform: rotation, non-uniform scale, shear, translation
        a = rng.standard_normal((3, 3)) * 0.35 + np.eye(3) * rng.uniform(0.5, 1.6)
        obj = np.eye(4); obj[:3, :3] = a; obj[:3, 3] = rng.uniform(-0.5, 0.5, 3)
    if rng.random() < 0.3:                      # the camera close to, or inside, the cloud
        dist_scale *= float(rng.choice([0.03, 0.12, 0.3]))
    aspect = w / h
    proj = None
    if kind == 1:      # off-centre frustum
        r = near / 2.41421
        cx, cy = rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4)
        proj = pkg.camera.frustum((cx - 1) * r, (cx + 1) * r, (cy - 1) * r / aspect, (cy + 1) * r / aspect, near, far)
    elif kind == 2:    # orthographic
        half = rng.uniform(0.6, 1.6)
        proj = pkg.camera.orthographic(-half, half, -half / aspect, half / aspect, near, far)
    return pkg.camera.make_camera(w, h, sh_order=order, frame=frame, distance=4.61995 * dist_scale, proj_matrix=proj, near=near, far=far, object_matrix=obj)


def make_script(rng, it, heavy, longlived):
    """everything an iteration does, drawn up front (the random stream does not depend on what the GPU answers)"""
    n = int(rng.choice([1000000, 1500000, 2500000])) if heavy else int(rng.choice([0, 1, 63, 64, 65, 1000, 20000, 100000, 400000]))
    sh = bool(rng.integers(0, 2))
    lo = rng.uniform(-6.0, -3.0)
    splats = pkg.scenes.make_scene(n, seed=int(rng.integers(1, 1 << 30)), sh=sh, log_scale_range=(lo, lo + rng.uniform(0.5, 2.5)))
    if n >= 1000 and rng.random() < 0.3:      # a few huge splats (the cooperative big-rect path)
        k = min(200, n)
        splats.scale[:k] = pkg.scenes.f16bits(rng.uniform(0.2, 1.5, size=(k, 3)))
    if n >= 1000 and rng.random() < 0.3:      # coincident splats: ties in the sort
        k = int(rng.integers(2, 80))
        splats.P[100:100 + k] = splats.P[100]
    if n >= 64 and rng.random() < 0.25:       # poisoned attributes: NaN, infinities, zeros, absurd magnitudes
        f32 = np.array([np.nan, np.inf, -np.inf, 1.0e30, -1.0e30, 0.0, 1.0e-30], np.float32)
        h16 = pkg.scenes.f16bits(np.array([np.nan, np.inf, -np.inf, 65504.0, 0.0, 6.0e-8, -1.0], np.float32))
        for _ in range(int(rng.integers(1, 12))):
            i = int(rng.integers(0, n))
            what = int(rng.integers(0, 5))
            if what == 0: splats.P[i, int(rng.integers(0, 3))] = rng.choice(f32)
            elif what == 1: splats.scale[i, int(rng.integers(0, 3))] = rng.choice(h16)
            elif what == 2: splats.orient[i, :] = rng.choice(h16) if rng.random() < 0.5 else pkg.scenes.f16bits(np.zeros(4, np.float32))
            elif what == 3: splats.alpha[i] = rng.choice(np.array([np.nan, np.inf, -1.0, 2.0, 0.0, 1.0 / 255.0, 1.0], np.float32))
            else: splats.Cd[i, int(rng.integers(0, 3))] = rng.choice(h16)
    w = 1920 if heavy else int(rng.choice([64, 333, 640, 1280, 1920, 2500, 2500, 5000, 9000, 16384]))
    h = 1080 if heavy else (int(rng.choice([48, 217])) if w > 2500 else int(rng.choice([48, 217, 480, 720, 1080])))
    order = int(rng.integers(0, 4)) if sh else 0
    kind = int(rng.integers(0, 3))
    count = int(rng.choice([1, 1, 2, 3, 8]))
    index = int(rng.integers(0, count))
    layout = int(rng.integers(0, 2))
    opts = {E.OPT_OCCLUSION_CULL: int(rng.choice([0, 1, 2, 3])), E.OPT_FRONT_SLAB: int(rng.choice([0, 1, 2])),
            E.OPT_LOCAL_SORT: int(rng.choice([0, 1, 2])), E.OPT_LAZY_COLOUR: int(rng.choice([0, 1, 2])),
            E.OPT_FRAMES_IN_FLIGHT: int(rng.choice([1, 1, 2])), E.OPT_CLUSTER_CULL: int(rng.choice([0, 1, 1])),
            E.OPT_STORAGE_ORDER: int(rng.choice([0, 1, 1])), E.OPT_XCD_SWIZZLE: int(rng.choice([0, 1, 2, 3])),
            E.OPT_CULL_DILATE: int(rng.choice([0, 1, 2, 5])), E.OPT_SORT_CACHE: int(rng.choice([0, 1, 1, 2])),
            E.OPT_DEFERRED_CHECK: int(rng.choice([0, 0, 0, 1])), E.OPT_SUPER_TILE: int(rng.choice([0, 0, 2, 8]))}
    if rng.random() < (0.7 if heavy else 0.25):     # the library as it comes
        opts = {}
    for kv in os.environ.get("FUZZ_FORCE", "").split():       # (debugging: FUZZ_FORCE="7=0 16=1" overrides options in every iteration)
        k_, v_ = kv.split("=")
        opts[int(k_)] = int(v_)
    use_depth = bool(rng.random() < 0.35)
    multi = int(rng.choice([0, 0, 0, 2, 3, 8])) if not (heavy or longlived) else 0
    if longlived:     # every option gets a definite value (the context remembers the last iteration's)
        full = dict(DEFAULTS); full.update(opts); opts = full
    if multi:
        index, count = 0, 1
        opts.pop(E.OPT_DEFERRED_CHECK, None)       # (gathered frames are always checked)
    path = [(0, 1.0), (1, 1.0), (2, 1.0), (40, 1.3), (41, 1.3), (41, 1.3), (3, 1.0), (4, 0.7)]
    cams = [random_camera(np.random.default_rng(1000 + it), w, h, order, f, d, kind) for f, d in path]
    if rng.random() < 0.3:       # the camera turning in place (the position-keyed sort cache skips those sorts)
        cams[2] = pkg.camera.rotated_in_place(cams[1], 7.0, -3.0)
        cams[5] = pkg.camera.rotated_in_place(cams[4], -11.0, 2.0)
    cuts = None
    if not multi and n >= 8 and rng.random() < 0.25:      # the same cloud staged as several entries (gsr_upload_begin / append / end)
        cuts = sorted(set(int(v) for v in rng.integers(1, n, size=int(rng.integers(1, 4)))))
    flips = {}
    if longlived:
        for k in range(len(cams)):
            if rng.random() < 0.3:       # an option flipped, or the shard changed, in mid-stream
                which = int(rng.integers(0, 9))
                if which < 8:
                    o, vals = FLIPS[which]
                    flips[k] = ("opt", o, int(rng.choice(vals)))
                else:
                    cnt2 = int(rng.choice([1, 2, 3, 8]))
                    flips[k] = ("shard", int(rng.integers(0, cnt2)), cnt2, int(rng.integers(0, 2)))
    dev_target = bool(rng.random() < 0.3) and not multi and not use_depth     # the frames go to a DEVICE buffer (no staging copy; deferred hand-over possible)
    user_stream = dev_target and bool(rng.random() < 0.6)                       # ... on the caller's own stream
    depth = depth_image(np.random.default_rng(it), w, h, cams[0], splats.P) if use_depth else None
    desc = dict(it=it, n=n, sh=sh, w=w, h=h, order=order, proj=kind, shard=(index, count, layout), depth=use_depth, multi=multi, parts=cuts, dev_target=dev_target, user_stream=user_stream,
                opts={int(k): v for k, v in opts.items()}, flips={k: v for k, v in flips.items()})
    return dict(splats=splats, n=n, w=w, h=h, shard=(index, count, layout), opts=opts, depth=depth, multi=multi, cams=cams, cuts=cuts, flips=flips, desc=desc, dev_target=dev_target, user_stream=user_stream)


def depth_image(rng, w, h, cam, P):
    """an opaque pass's depth buffer: per-pixel noise (the only kind before round 6: every tile's largest depth is 1, nothing for K1 to
    cull), the far plane, blocks of tile size and of odd sizes whose values sit INSIDE the cloud's range of window depths (and NaN, 0,
    negative, beyond 1), a horizontal ramp through that range, a disc"""
    M = np.asarray(cam.proj, np.float64).reshape(4, 4).T @ np.asarray(cam.obj_view, np.float64).reshape(4, 4).T
    Ph = np.concatenate([np.nan_to_num(P[:5000].astype(np.float64)), np.ones((min(len(P), 5000), 1))], axis=1) @ M.T
    ok = (Ph[:, 3] > 0) & (np.abs(Ph[:, 2]) <= Ph[:, 3])
    zw = 0.5 * Ph[ok, 2] / Ph[ok, 3] + 0.5 if ok.any() else np.array([0.5])
    q = [float(np.quantile(zw, x)) for x in (0.05, 0.25, 0.5, 0.75, 0.95)]
    kind = int(rng.integers(0, 6))
    if kind == 0:
        return np.where(rng.random((h, w)) < 0.5, np.float32(q[2]), np.float32(1.0)).astype(np.float32)
    if kind == 1:
        return np.ones((h, w), np.float32)
    if kind in (2, 3):
        B = 16 if kind == 2 else int(rng.choice([5, 40, 128, 300]))
        vals = np.asarray(q + [1.0, 1.0, 1.0] + ([0.0, 1.5, -0.5, np.nan] if rng.random() < 0.4 else []), np.float32)
        g = vals[rng.integers(0, len(vals), ((h + B - 1) // B, (w + B - 1) // B))]
        return np.ascontiguousarray(g.repeat(B, 0).repeat(B, 1)[:h, :w])
    if kind == 4:
        return np.ascontiguousarray(np.broadcast_to((q[0] + (q[4] - q[0]) * np.arange(w) / max(w - 1, 1)).astype(np.float32), (h, w)))
    yy, xx = np.mgrid[0:h, 0:w]
    r2 = ((xx - w / 2) / (0.3 * w)) ** 2 + ((yy - h / 2) / (0.3 * h)) ** 2
    return np.where(r2 < 1.0, q[1] + (q[3] - q[1]) * r2, 1.0).astype(np.float32)


def execute(sc, dut, verbose=False):
    """run one script on `dut` (and a fresh plain context): (frames compared, None) or (frames, text of the first difference)"""
    splats, n, multi, opts, depth = sc["splats"], sc["n"], sc["multi"], sc["opts"], sc["depth"]
    index, count, layout = sc["shard"]
    plain = E.Engine(0)
    frames = 0
    try:
        if multi:
            dut.set_option(E.OPT_SHARD_LAYOUT, layout)
        for e in ((plain,) if multi else (dut, plain)):
            e.set_option(E.OPT_SHARD_LAYOUT, layout)
            e.set_row_shard(index, count)
        # (the storage order is a property of the product under test: ties are drawn in storage order, so the plain context stores alike)
        plain.set_option(E.OPT_STORAGE_ORDER, opts.get(E.OPT_STORAGE_ORDER, 1))
        plain.set_option(E.OPT_OCCLUSION_CULL, 0); plain.set_option(E.OPT_CLUSTER_CULL, 0)
        plain.set_option(E.OPT_LOCAL_SORT, 0); plain.set_option(E.OPT_LAZY_COLOUR, 0)
        plain.set_option(E.OPT_DEBUG_FLAGS, 32)      # (GSR_FLAG_NO_DEPTH_CLASS: depth-tested frames stage every record and compare every fragment)
        for k, v in opts.items():
            dut.set_option(k, v)
        if sc["cuts"]:
            S = pkg.scenes.Splats
            cuts = sc["cuts"]
            parts = [S(P=splats.P[a:b], Cd=splats.Cd[a:b], alpha=splats.alpha[a:b], scale=splats.scale[a:b], orient=splats.orient[a:b],
                       shx=None if splats.shx is None else splats.shx[a:b], shy=None if splats.shy is None else splats.shy[a:b],
                       shz=None if splats.shz is None else splats.shz[a:b]) for a, b in zip([0] + cuts, cuts + [n])]
            dut.upload_parts(parts)
        else:
            dut.upload(splats)
        plain.upload(splats)
        count_now = count
        devbuf = [C.c_void_p(), 0]
        ustream = [C.c_void_p()]
        deferred = bool(opts.get(E.OPT_DEFERRED_CHECK, 0)) and not multi
        truncated_seen = dut.stats()["frames_truncated"] if not multi else 0
        for k, c in enumerate(sc["cams"]):
            fl = sc["flips"].get(k)
            if fl is not None:
                if fl[0] == "opt":
                    dut.set_option(fl[1], fl[2])
                else:
                    for e in (dut, plain):
                        e.set_option(E.OPT_SHARD_LAYOUT, fl[3])
                        e.set_row_shard(fl[1], fl[2])
                    count_now = fl[2]
            want = plain.render(c) if depth is None else plain.render_depth(c, depth)
            if sc.get("dev_target"):
                rows_ = dut.band_rows(c.height) if count_now > 1 else c.height
                nbytes = rows_ * c.width * 16
                if devbuf[1] < nbytes:
                    if devbuf[0].value: HIP.hipFree(devbuf[0])
                    assert HIP.hipMalloc(C.byref(devbuf[0]), C.c_size_t(nbytes)) == 0
                    devbuf[1] = nbytes
                if sc.get("user_stream"):
                    # the caller's own stream (what bench.py does): the clear is QUEUED on it in front of the frame, the read-back behind
                    # it -- no host synchronisation in between; the library has to order its kernels against both
                    if not ustream[0].value:
                        assert HIP.hipStreamCreate(C.byref(ustream[0])) == 0
                        dut.set_stream(ustream[0].value)
                    assert HIP.hipMemsetAsync(devbuf[0], 0xff if k % 2 else 0, C.c_size_t(nbytes), ustream[0]) == 0      # (poison, then zeros: the
                    assert HIP.hipMemsetAsync(devbuf[0], 0, C.c_size_t(nbytes), ustream[0]) == 0                          #  frame must come after both)
                    dut.render_to_device(c, devbuf[0].value)
                    got = np.empty((rows_, c.width, 4), np.float32)
                    assert HIP.hipMemcpyAsync(C.c_void_p(got.ctypes.data), devbuf[0], C.c_size_t(nbytes), 2, ustream[0]) == 0
                    assert HIP.hipStreamSynchronize(ustream[0]) == 0
                else:
                    assert HIP.hipMemset(devbuf[0], 0, C.c_size_t(nbytes)) == 0      # (band padding is never written in a device target)
                    dut.render_to_device(c, devbuf[0].value)
                    dut.synchronize()
                    got = np.empty((rows_, c.width, 4), np.float32)
                    assert HIP.hipMemcpy(C.c_void_p(got.ctypes.data), devbuf[0], C.c_size_t(nbytes), 2) == 0
            else:
                got = (dut.render(c) if depth is None else (dut.render(c, depth) if multi else dut.render_depth(c, depth)))
            frames += 1
            if deferred:
                # (deferred hand-over: a frame whose lists outgrew the buffer is handed over with clamped lists and COUNTED --
                #  the documented price of never waiting for the pair count; only the frames that were not truncated are exact)
                tr = dut.stats()["frames_truncated"]
                if tr != truncated_seen:
                    truncated_seen = tr
                    continue
            if not np.array_equal(got, want, equal_nan=True):
                d = np.nan_to_num(np.abs(got - want), nan=1.0)
                rows = np.nonzero(d.max(axis=(1, 2)))[0]
                st = dut.stats()
                return frames, (f"frame {k}: max |diff| {float(d.max())}, {int((d.max(axis=-1) > 0).sum())} pixels, rows {rows[0]}..{rows[-1]} | "
                                + str({q: st[q] for q in ("frames_culled", "frames_slab", "frames_jumped", "frames_repaired", "frames_resorted", "frames_requeued", "n_visible", "pairs_total")}))
        if verbose:
            st = dut.stats()
            print("ok", sc["desc"]["it"], "n", n, f"{sc['w']}x{sc['h']}", "shard", sc["shard"], "multi", multi, "depth", int(depth is not None),
                  "cull", opts.get(E.OPT_OCCLUSION_CULL, "-"), "slab", opts.get(E.OPT_FRONT_SLAB, "-"),
                  "| culled", st["frames_culled"], "slab", st["frames_slab"], "jumped", st["frames_jumped"], "repaired", st["frames_repaired"],
                  "resorted", st["frames_resorted"], flush=True)
        return frames, None
    finally:
        plain.close()
        try:
            if ustream[0].value:
                dut.set_stream(0)
                HIP.hipStreamDestroy(ustream[0])
            if devbuf[0].value: HIP.hipFree(devbuf[0])
        except NameError:
            pass


def fresh_dut(sc):
    return E.MultiEngine([0] * sc["multi"], E.TRANSPORT_COPY) if sc["multi"] else E.Engine(0)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # (skip the iterations before this one: same random sequence)
    mode = sys.argv[4] if len(sys.argv) > 4 else "0"
    heavy, longlived = mode == "1", mode in ("2", "3")
    multi_ll = 3 if mode == "3" else 0         # mode 3: the long-lived context is a gsr_multi of three contexts on this GPU
    rng = np.random.default_rng(seed)
    t_start = time.time()
    frames = 0
    keep_dut = None
    history = []
    for it in range(iters):
        sc = make_script(rng, it, heavy, longlived)
        if multi_ll:
            sc["multi"] = multi_ll; sc["shard"] = (0, 1, sc["shard"][2]); sc["cuts"] = None; sc["dev_target"] = False; sc["user_stream"] = False
            sc["opts"].pop(E.OPT_DEFERRED_CHECK, None)
            sc["flips"] = {k: v for k, v in sc["flips"].items() if v[0] == "opt"}
            sc["desc"].update(multi=multi_ll, shard=sc["shard"], parts=None, dev_target=False, flips=sc["flips"])
        if it < first:
            continue
        print("..", sc["desc"], flush=True)
        history = (history + [sc])[-4:] if longlived else [sc]
        if longlived:
            if keep_dut is None:
                keep_dut = E.MultiEngine([0] * multi_ll, E.TRANSPORT_COPY) if multi_ll else E.Engine(0)
            dut = keep_dut
        else:
            dut = fresh_dut(sc)
        try:
            nf, bad = execute(sc, dut, verbose=True)
        finally:
            if not longlived:
                dut.close()
        frames += nf
        if bad:
            print("MISMATCH", bad, sc["desc"], flush=True)
            # does the context's history matter?  the iteration alone on a fresh context, then behind its predecessors
            for depth_ in range(1, len(history) + 1):
                d2 = fresh_dut(sc)
                try:
                    res = None
                    for s2 in history[-depth_:]:
                        _, res = execute(s2, d2)
                finally:
                    d2.close()
                print(f"   replayed on a fresh context behind {depth_ - 1} predecessor(s):", "REPRODUCED " + res if res else "no difference", flush=True)
                if res:
                    for o in sorted(sc["opts"]):      # which options of the failing iteration matter?
                        if sc["opts"][o] == DEFAULTS.get(o):
                            continue
                        alt = dict(sc); alt["opts"] = dict(sc["opts"]); alt["opts"][o] = DEFAULTS[o]
                        d3 = fresh_dut(sc)
                        try:
                            r3 = None
                            for s2 in history[-depth_:-1] + [alt]:
                                _, r3 = execute(s2, d3)
                        finally:
                            d3.close()
                        print(f"      with option {o} at its default {DEFAULTS[o]}:", "still differs" if r3 else "NO difference", flush=True)
                    break
            return 1
    print(f"{iters} iterations, {frames} frames bit-identical, {time.time() - t_start:.0f} s")
    return 0


if __name__ == "__main__":
    sys.exit(main())

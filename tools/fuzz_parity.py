"""Randomised exactness soak (GPU box): python tools/fuzz_parity.py [iterations] [seed] [first] [heavy]
Every iteration: a random cloud (0 .. 400 k splats, random scale range, SH or not), a random framebuffer, a random projection
(perspective / off-centre / orthographic), a random row shard and random library options (occlusion culling mode, front slab,
small-frame sort, lazy colour, frames in flight, cluster culling, storage order), then a short camera path (small steps, a
jump, a repeat) -- every frame must be BIT-IDENTICAL to the same camera from a context that culls nothing, takes the global
sort and shades eagerly (that configuration is what the -m gpu tests hold against the CPU oracle).
Some iterations render depth-tested (an opaque pass's depth image in front of part of the frame), some go through gsr_multi_*
(several contexts on this GPU, COPY transport: shard, render, gather) and are compared with the unsharded frame.
heavy = 1: clouds of 1 - 2.5 M splats at 1920x1080 (the policy's temporal culling and front-slab frames engage by themselves).
heavy = 2: ONE long-lived context under test for the whole run (re-uploads, option flips, shard and shape changes between iterations).
Exits non-zero at the first difference, printing the configuration that produced it."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
E = pkg.engine


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


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # (skip the iterations before this one: same random sequence)
    heavy = len(sys.argv) > 4 and sys.argv[4] == "1"
    longlived = len(sys.argv) > 4 and sys.argv[4] == "2"      # ONE context under test for the whole run: re-uploads, option flips, shape changes
    keep_dut = None
    rng = np.random.default_rng(seed)
    t_start = time.time()
    frames = 0
    for it in range(iters):
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
                E.OPT_CULL_DILATE: int(rng.choice([0, 1, 2, 5])), E.OPT_SORT_CACHE: int(rng.choice([0, 1, 1])),
                E.OPT_SUPER_TILE: int(rng.choice([0, 0, 2, 8]))}
        if rng.random() < (0.7 if heavy else 0.25):     # the library as it comes
            opts = {}
        use_depth = rng.random() < 0.25
        multi = int(rng.choice([0, 0, 0, 2, 3, 8])) if not (heavy or longlived) else 0
        if longlived:     # every option gets a definite value (the context remembers the last iteration's)
            full = {E.OPT_OCCLUSION_CULL: 1, E.OPT_FRONT_SLAB: 1, E.OPT_LOCAL_SORT: 1, E.OPT_LAZY_COLOUR: 1, E.OPT_FRAMES_IN_FLIGHT: 1, E.OPT_CLUSTER_CULL: 1,
                    E.OPT_STORAGE_ORDER: 1, E.OPT_XCD_SWIZZLE: 2, E.OPT_CULL_DILATE: 2, E.OPT_SORT_CACHE: 1, E.OPT_SUPER_TILE: 0}
            full.update(opts)
            opts = full
        if multi:
            index, count = 0, 1
        desc = dict(it=it, n=n, sh=sh, w=w, h=h, order=order, proj=kind, shard=(index, count, layout), depth=use_depth, multi=multi,
                    opts={int(k): v for k, v in opts.items()})
        depth = None
        if use_depth:
            depth = np.where(np.random.default_rng(it).random((h, w)) < 0.5, 0.5, 1.0).astype(np.float32)
        path = [(0, 1.0), (1, 1.0), (2, 1.0), (40, 1.3), (41, 1.3), (41, 1.3), (3, 1.0), (4, 0.7)]
        cams = [random_camera(np.random.default_rng(1000 + it), w, h, order, f, d, kind) for f, d in path]
        if it < first:
            continue
        print("..", desc, flush=True)
        if longlived:
            if keep_dut is None:
                keep_dut = E.Engine(0)
            dut, plain = keep_dut, E.Engine(0)
        else:
            dut, plain = (E.MultiEngine([0] * multi, E.TRANSPORT_COPY) if multi else E.Engine(0)), E.Engine(0)
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
            for k, v in opts.items():
                dut.set_option(k, v)
            if not multi and n >= 8 and rng.random() < 0.25:      # the same cloud staged as several entries (gsr_upload_begin / append / end)
                cuts = sorted(set(int(v) for v in rng.integers(1, n, size=int(rng.integers(1, 4)))))
                S = pkg.scenes.Splats
                parts = []
                for a, b in zip([0] + cuts, cuts + [n]):
                    parts.append(S(P=splats.P[a:b], Cd=splats.Cd[a:b], alpha=splats.alpha[a:b], scale=splats.scale[a:b], orient=splats.orient[a:b],
                                   shx=None if splats.shx is None else splats.shx[a:b], shy=None if splats.shy is None else splats.shy[a:b],
                                   shz=None if splats.shz is None else splats.shz[a:b]))
                dut.upload_parts(parts)
            else:
                dut.upload(splats)
            plain.upload(splats)
            for k, c in enumerate(cams):
                if longlived and rng.random() < 0.3:       # an option flipped, or the shard changed, in mid-stream
                    which = int(rng.integers(0, 9))
                    choices = [(E.OPT_OCCLUSION_CULL, [0, 1, 2, 3]), (E.OPT_FRONT_SLAB, [0, 1, 2]), (E.OPT_LOCAL_SORT, [0, 1, 2]), (E.OPT_LAZY_COLOUR, [0, 1, 2]),
                               (E.OPT_FRAMES_IN_FLIGHT, [1, 2]), (E.OPT_CLUSTER_CULL, [0, 1]), (E.OPT_XCD_SWIZZLE, [0, 1, 2, 3]), (E.OPT_CULL_DILATE, [0, 1, 2, 5])]
                    if which < 8:
                        o, vals = choices[which]
                        dut.set_option(o, int(rng.choice(vals)))
                    else:
                        cnt2 = int(rng.choice([1, 2, 3, 8])); idx2 = int(rng.integers(0, cnt2)); lay2 = int(rng.integers(0, 2))
                        for e in (dut, plain):
                            e.set_option(E.OPT_SHARD_LAYOUT, lay2)
                            e.set_row_shard(idx2, cnt2)
                want = plain.render(c) if depth is None else plain.render_depth(c, depth)
                got = (dut.render(c) if depth is None else (dut.render(c, depth) if multi else dut.render_depth(c, depth)))
                frames += 1
                if not np.array_equal(got, want, equal_nan=True):
                    d = np.abs(got - want)
                    print("MISMATCH frame", k, "max |diff|", float(np.nanmax(d)), "pixels", int((d.max(axis=-1) > 0).sum()), desc, dut.stats())
                    return 1
            st = dut.stats()
            print("ok", it, "n", n, f"{w}x{h}", "proj", kind, "shard", (index, count, layout), "multi", multi, "depth", int(use_depth), "cull", opts.get(E.OPT_OCCLUSION_CULL, "-"), "slab", opts.get(E.OPT_FRONT_SLAB, "-"),
                  "| culled", st["frames_culled"], "slab", st["frames_slab"], "jumped", st["frames_jumped"], "repaired", st["frames_repaired"], "resorted", st["frames_resorted"], flush=True)
        finally:
            if not longlived:
                dut.close()
            plain.close()
    print(f"{iters} iterations, {frames} frames bit-identical, {time.time() - t_start:.0f} s")
    return 0


if __name__ == "__main__":
    sys.exit(main())

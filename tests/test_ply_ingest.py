"""CPU test of the INRIA PLY ingest (SURVEY N1 / App. D conventions)."""
import numpy as np


def _write_ply(path, n, rng, binary=True):
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{k}" for k in range(45)] + \
            ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    data = rng.normal(0, 1, (n, len(names))).astype(np.float32)
    hdr = "ply\nformat %s 1.0\nelement vertex %d\n" % ("binary_little_endian" if binary else "ascii", n)
    hdr += "".join(f"property float {nm}\n" for nm in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if binary:
            f.write(data.tobytes())
        else:
            for row in data:
                f.write((" ".join(repr(float(x)) for x in row) + "\n").encode())
    return names, data


def test_inria_ply_round_trip(pkg, oracle, tmp_path):
    rng = np.random.default_rng(3)
    for binary in (True, False):
        path = str(tmp_path / ("a.ply" if binary else "b.ply"))
        names, d = _write_ply(path, 257, rng, binary)
        col = {nm: d[:, k] for k, nm in enumerate(names)}
        s = pkg.ply.load_inria_ply(path)
        assert s.n == 257 and s.has_sh
        assert np.array_equal(s.P, np.stack([col["x"], col["y"], col["z"]], 1))
        h = lambda a: np.asarray(a, np.float32).astype(np.float16).view(np.uint16)
        assert np.array_equal(s.Cd, h(0.28209479177387814 * np.stack([col["f_dc_0"], col["f_dc_1"], col["f_dc_2"]], 1) + 0.5))
        assert np.allclose(s.alpha, 1 / (1 + np.exp(-col["opacity"].astype(np.float64))), rtol=1e-6)
        assert np.array_equal(s.scale, h(np.exp(np.stack([col["scale_0"], col["scale_1"], col["scale_2"]], 1))))
        q = np.stack([col["rot_1"], col["rot_2"], col["rot_3"], col["rot_0"]], 1)   # (x, y, z, w)
        assert np.array_equal(s.orient, h(q / np.linalg.norm(q, axis=1, keepdims=True)))
        for k in range(15):   # sh{k+1} = (f_rest_k, f_rest_{k+15}, f_rest_{k+30})
            assert np.array_equal(s.shx[:, k], h(col[f"f_rest_{k}"]))
            assert np.array_equal(s.shy[:, k], h(col[f"f_rest_{k + 15}"]))
            assert np.array_equal(s.shz[:, k], h(col[f"f_rest_{k + 30}"]))
        assert (s.shx[:, 15] == 0).all()
    # and it renders through the oracle (smoke: finite, non-empty)
    cam = pkg.camera.make_camera(64, 48, sh_order=3)
    img = oracle.render(pkg.ply.load_inria_ply(path, cd_override=(0.5, 0.5, 0.5)), cam)
    assert np.isfinite(img).all() and img[..., 3].max() > 0


def test_ply_door_of_the_bench(pkg, tmp_path):
    """bench.py --ply PATH: the capture becomes a config of its own (kind "ply") through the example scene's activations, with an
    orbit fitted to the cloud -- pivot at the median position, the camera outside the radius that holds 80 % of the points and
    looking at the pivot"""
    v = pkg.scenes.make_inria_raw(5000, seed=4, radius=0.7)
    v["x"] += 3.0; v["y"] -= 1.0                                  # a cloud that is NOT about the origin
    v["x"][:50] += 400.0                                          # ... with far-away background points, as captures have
    path = str(tmp_path / "capture.ply")
    pkg.scenes.write_inria_ply(path, v)
    name = pkg.scenes.register_ply_config(path, pkg.ply, name="PLY_TEST")
    try:
        splats, cfg = pkg.scenes.make_config(name)
        ref = pkg.ply.load_inria_ply(path)
        assert splats.n == 5000 and cfg["kind"] == "ply" and cfg["sh_order"] == 3
        for a, b in ((splats.P, ref.P), (splats.Cd, ref.Cd), (splats.alpha, ref.alpha), (splats.scale, ref.scale), (splats.orient, ref.orient), (splats.shx, ref.shx)):
            assert np.array_equal(a, b)
        assert np.allclose(cfg["pivot"], (3.0, -1.0, 0.0), atol=0.08)
        assert 0.5 < cfg["distance"] < 2.0                       # (1.6 x the 80 % radius of a 0.7 ball: ~1.0; the 400-unit outliers do not stretch it)
        cam = pkg.scenes.config_camera(name, pkg.camera, cfg["width"], cfg["height"], cfg["sh_order"], 7)
        assert abs(np.linalg.norm(cam.cam_pos - np.asarray(cfg["pivot"], np.float32)) - cfg["distance"]) < 1e-3
        view = cam.view.reshape(4, 4).T
        pv = view @ np.array([*cfg["pivot"], 1.0])
        assert abs(pv[0]) < 1e-3 and abs(pv[1]) < 1e-3 and pv[2] < 0     # the pivot is dead ahead
        sub, _ = pkg.scenes.make_config(name, 100)
        assert sub.n == 100
    finally:
        pkg.scenes.CONFIGS.pop("PLY_TEST", None)
    import bench
    src = open(bench.__file__).read()
    assert "--ply" in src and src.count("register_ply_config") == 2          # both entry forms (one process / one process per GPU)


def test_capture_shaped_generator(pkg, oracle):
    """R1 (scenes.make_capture): reproducible; surfaces + 30 % near-transparent floaters + a heavy tail of sizes + first-order SH that
    dominates; and from inside the room every ray ends on a surface (the oracle's frame is opaque everywhere)"""
    a = pkg.scenes.make_capture(40000, seed=5)
    b = pkg.scenes.make_capture(40000, seed=5)
    assert a.n == 40000 and np.array_equal(a.P, b.P) and np.array_equal(a.scale, b.scale) and np.array_equal(a.shx, b.shx)
    op = a.alpha
    assert 0.25 < (op < 0.2).mean() < 0.40                       # the floaters
    sc = a.scale.view(np.float16).astype(np.float32)
    big = sc.max(axis=1)
    assert np.quantile(big, 0.5) < 0.03 and (big > 0.3).sum() >= 100      # a few hundred splats fill a good part of the screen
    flat = sc.min(axis=1) / sc.max(axis=1)
    assert np.median(flat) < 0.35                                # surface splats are flat
    sh = a.shx.view(np.float16).astype(np.float32)
    assert np.abs(sh[:, :3]).mean() > 2.0 * np.abs(sh[:, 3:15]).mean()
    q = a.orient.view(np.float16).astype(np.float32)
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=2e-3)
    cam = pkg.scenes.config_camera("R1", pkg.camera, 160, 90, 3, 2)
    img = oracle.render(a, cam)
    assert np.isfinite(img).all() and (img[..., 3] > 0.9).mean() > 0.9

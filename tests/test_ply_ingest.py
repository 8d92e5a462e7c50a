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

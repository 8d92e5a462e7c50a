"""How unevenly a scene loads its tiles: list entries scanned, hits gathered and wave-record evaluations per tile.
   GPU box: python tools/tile_hits.py T1 S1 C4"""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
for name in sys.argv[1:]:
    splats, cfg = pkg.scenes.make_config(name)
    W, H, order = cfg["width"], cfg["height"], cfg["sh_order"]
    eng = pkg.Engine(0)
    eng.upload(splats)
    for f in range(3):
        eng.render(pkg.scenes.config_camera(name, pkg.camera, W, H, order, f))
    tw = eng.debug_tile_work().reshape(-1, 4).astype(np.int64)
    for k, lab in ((0, "entries scanned"), (1, "hits gathered"), (2, "wave evaluations")):
        v = tw[:, k]
        print("%s %-17s sum %9.0f  median %6.0f  p90 %6.0f  p99 %6.0f  max %7.0f   tiles above 4x median: %d, their share of the sum %.2f" % (
            name, lab, v.sum(), *np.quantile(v, [0.5, 0.9, 0.99, 1.0]), int((v > 4 * np.median(v)).sum()), v[v > 4 * np.median(v)].sum() / max(1, v.sum())))
    sat = (tw[:, 3] & 1).astype(bool)
    print("%s tiles %d, went opaque %d, drew something %d" % (name, len(tw), int(sat.sum()), int((tw[:, 1] > 0).sum())))
    del eng

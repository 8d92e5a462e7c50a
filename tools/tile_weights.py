"""Per-config tile-work distribution (what k_sum_work's heaviest-first verdict sees).  GPU box: python tools/tile_weights.py"""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
for name in (sys.argv[1:] or ["C1", "C2", "C3", "C4", "C5"]):
    splats, cfg = pkg.scenes.make_config(name)
    eng = pkg.Engine(0)
    eng.upload(splats)
    for f in range(3):
        eng.render(pkg.camera.make_camera(cfg["width"], cfg["height"], sh_order=cfg.get("sh_order", 3), frame=f))
    tw = eng.debug_tile_work().reshape(-1, 4).astype(np.int64)
    w = tw[:, 2] * 32 + tw[:, 1] * 8 + (tw[:, 0] >> 1)
    W, wmax = w.sum(), w.max()
    print("%s tiles %d W %.1f M instr (balanced %.0f us at 1.05 ns/1024 SIMDs) wmax %.0f k  wmax*1536/W %.2f  p50 %.0f p90 %.0f p99 %.0f k"
          % (name, len(w), W / 1e6, W * 1.05e-3 / 1024, wmax / 1e3, wmax * 1536 / max(W, 1), *(np.quantile(w, [0.5, 0.9, 0.99]) / 1e3)))
    del eng

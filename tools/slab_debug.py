"""debug: front-slab frames against plain frames (where do they differ?)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
E = pkg.engine
splats = pkg.scenes.make_scene(400000, seed=191, sh=True, radius=1.0)
w, h = 960, 540
cams = [pkg.camera.make_camera(w, h, sh_order=3, frame=i) for i in (0, 1, 2, 3, 4, 40, 41, 42)]
cams += [pkg.camera.make_camera(w, h, sh_order=3, frame=43, distance=d) for d in (2.2, 2.25, 6.0, 5.9)]
plain, eng = pkg.Engine(0), pkg.Engine(0)
plain.set_option(E.OPT_OCCLUSION_CULL, 0)
plain.upload(splats); eng.upload(splats)
for mode, slab in ((3, 1), (2, 1), (2, 0)):
    eng.set_option(E.OPT_OCCLUSION_CULL, mode)
    eng.set_option(E.OPT_FRONT_SLAB, slab)
    eng.upload(splats)
    eng.stats_reset()
    print("mode", mode, "front slab", slab)
    for k, c in enumerate(cams):
        a, b = eng.render(c), plain.render(c)
        st = eng.stats()
        d = np.abs(a - b).max(axis=2)
        bad = d > 0
        msg = ""
        if bad.any():
            ys, xs = np.nonzero(bad)
            tiles = sorted(set(zip((ys // 16).tolist(), (xs // 16).tolist())))
            msg = f"DIFF px={int(bad.sum())} max={d.max():.3e} tiles={len(tiles)} first tiles={tiles[:6]} rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()}"
        print(f"  frame {k}: slab={st['frames_slab']} culled={st['frames_culled']} repaired={st['frames_repaired']} nvis={st['n_visible']} pairs={st['pairs_total']} {msg}")

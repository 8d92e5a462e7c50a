"""How much of a one-pass frame's depth order is ever LOOKED at?  (the round-5 verdict's item 6: "sort less, not faster")
Per super-tile list: the deepest prefix any of its tiles scanned (k_blend's bookkeeping: entries read, incl. the prefetched step) over
the list's length; summed over the frame = the share of the sorted, binned pairs that a perfect "nearest buckets first, extend on
demand" scheme would still have to sort and bin.   python tools/list_prefix_probe.py [config ...]   (GPU box)"""
import sys
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
E = pkg.engine
for name in (sys.argv[1:] or ["C4", "T1", "S1", "R1"]):
    splats, cfg = pkg.scenes.make_config(name)
    W, H, order = cfg["width"], cfg["height"], cfg["sh_order"]
    eng = pkg.Engine(0)
    eng.set_option(E.OPT_OCCLUSION_CULL, 0)          # one-pass frames
    eng.upload(splats)
    for i in range(4):
        eng.render(pkg.scenes.config_camera(name, pkg.camera, W, H, order, i))
    st = eng.stats()
    ts, te, _ = eng.debug_tile_lists()
    work = eng.debug_tile_work()                      # [tiles_y, tiles_x, 4]: entries scanned, records gathered, evaluations, flags
    S = st["super_tile"] if "super_tile" in st else (st["tiles_x"] + st["stiles_x"] - 1) // st["stiles_x"]
    ty, tx = work.shape[:2]
    need = np.zeros(len(ts), np.int64)
    for y in range(ty):
        for x in range(tx):
            s_ = (y // S) * st["stiles_x"] + (x // S)
            need[s_] = max(need[s_], int(work[y, x, 0]))
    ln = (te - ts).astype(np.int64)
    need = np.minimum(need, ln)
    sat = (work[:, :, 3] & 1).mean()
    print("%s one pass: %d pairs in %d lists; deepest prefix any tile of a list read: %.1f %% of the pairs (median list %.1f %%, p90 %.1f %%); "
          "tiles that went opaque %.2f; mean share of its list a tile read %.1f %%" % (
              name, ln.sum(), (ln > 0).sum(), 100.0 * need.sum() / max(ln.sum(), 1), 100.0 * np.median(need[ln > 0] / ln[ln > 0]),
              100.0 * np.quantile(need[ln > 0] / ln[ln > 0], 0.9), sat,
              100.0 * np.mean([work[y, x, 0] / max(ln[(y // S) * st["stiles_x"] + (x // S)], 1) for y in range(ty) for x in range(tx)])))
    eng.close()

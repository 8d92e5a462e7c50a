#!/usr/bin/env python3
"""Offline look at the depth horizons (occlusion culling): renders two consecutive UNCULLED frames of a config, rebuilds the
per-tile horizons of the first from the blend kernel's bookkeeping exactly as k_tile_pass / k_horizon_dilate do, and reports
which tiles of the second frame would have broken them, and what those tiles look like.
    python tools/horizon_probe.py C5 [frame] [dilate]            (GPU box)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C5"
frame = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dil = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pkg = ge.load_package()
splats, cfg = pkg.scenes.make_config(cfgname)
W, H, order = cfg["width"], cfg["height"], cfg["sh_order"]
eng = pkg.Engine(0)
eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
eng.upload(splats)


def frame_data(i):
    cam = pkg.scenes.config_camera(cfgname, pkg.camera, W, H, order, i)
    eng.render(cam)
    st = eng.stats()
    tw = eng.debug_tile_work().astype(np.int64)
    ls, le, pv = eng.debug_tile_lists()
    key = eng.debug_records(splats.n)["key"]
    return st, tw, ls.astype(np.int64), le.astype(np.int64), pv, key


st0, tw0, ls0, le0, pv0, key0 = frame_data(frame)
st1, tw1, ls1, le1, pv1, key1 = frame_data(frame + 1)
S, sx = st0["super_tile"], st0["stiles_x"]
ty, tx = np.mgrid[0:tw0.shape[0], 0:tw0.shape[1]]
sti = (ty // S) * sx + (tx // S)
for slack_mode in ("scan", "own"):
    s0, ln = ls0[sti], (le0 - ls0)[sti]
    def extent(tw):
        es = tw[..., 3] >> 16
        return np.where((es == 0xffff) | (es * 1024 > tw[..., 0]) | (slack_mode == "scan"), tw[..., 0], es * 1024)
    rd, hits, sat = extent(tw0), tw0[..., 1], tw0[..., 3] & 1
    first = ((tw0[..., 3] >> 1) & 0x1fff) * 1024 if slack_mode == "own" else 0
    want = rd + (np.maximum(rd - first, 0) >> 2) + 1024
    ok = (sat == 1) & (want < ln)
    h = np.full(rd.shape, np.inf, np.float32)
    h[ok] = key0[pv0[(s0 + want)[ok]]]
    # dilate
    D = h.copy()
    for dy in range(-dil, dil + 1):
        for dx in range(-dil, dil + 1):
            sh = np.full_like(h, 0)
            ys, yd = (slice(max(dy, 0), h.shape[0] + min(dy, 0)), slice(max(-dy, 0), h.shape[0] + min(-dy, 0)))
            xs, xd = (slice(max(dx, 0), h.shape[1] + min(dx, 0)), slice(max(-dx, 0), h.shape[1] + min(-dx, 0)))
            sh[yd, xd] = h[ys, xs]
            D = np.maximum(D, sh)
    s1, ln1 = ls1[sti], (le1 - ls1)[sti]
    rd1, sat1 = extent(tw1), tw1[..., 3] & 1
    kl = np.zeros(rd1.shape, np.float32)
    m = (sat1 == 1) & (rd1 > 0)
    kl[m] = key1[pv1[(s1 + rd1 - 1)[m]]]
    fin = np.isfinite(D)
    viol_open = fin & (sat1 == 0)
    viol_deep = fin & (sat1 == 1) & (kl > D)
    print(f"[{cfgname} frames {frame}->{frame + 1}, dilate {dil}, slack '{slack_mode}'] tiles {rd.size}, opaque {int(sat.sum())}, with horizon {int(np.isfinite(h).sum())}, "
          f"after dilation {int(fin.sum())};  would break: {int(viol_open.sum())} stayed open, {int(viol_deep.sum())} looked deeper")
    # what the horizons would keep: splats in front of the horizon of some tile of their rect (approximated by list membership)
    for name, v in (("open", viol_open), ("deeper", viol_deep)):
        if v.any():
            yy, xx = np.nonzero(v)
            print(f"   {name}: frame-0 hits of those tiles: median {np.median(hits[v]):.0f} (all opaque tiles: {np.median(hits[sat == 1]):.0f}); "
                  f"frame-0 rd/len median {np.median(rd[v] / np.maximum(ln[v], 1)):.2f}; frame-1 rd/len {np.median(rd1[v] / np.maximum(ln1[v], 1)):.2f}; "
                  + (f"distance to the nearest tile without horizon (tiles): "
                     f"{np.median([np.min(np.hypot(*(np.argwhere(~np.isfinite(h)) - [y, x]).T)) for y, x in list(zip(yy, xx))[:50]]):.1f}"
                     if (~np.isfinite(h)).any() else "every tile has a horizon"))
    if fin.any():
        frac = float((rd[sat == 1] / np.maximum(ln[sat == 1], 1)).mean())
        print(f"   opaque tiles scan on average {frac:.3f} of their list; horizon position / list length: median {np.median((want / np.maximum(ln, 1))[ok]):.3f}")
eng.close()

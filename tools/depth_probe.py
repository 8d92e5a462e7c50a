"""GPU box: frame by frame, what a depth-tested orbit does to the policies (python tools/depth_probe.py [C4] [far|occluder] [frames] [flags])"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
E = pkg.engine
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
kind = sys.argv[2] if len(sys.argv) > 2 else "occluder"
nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 40
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sync = (sys.argv[5] if len(sys.argv) > 5 else "sync") == "sync"      # "async": back-to-back frames, stats only (gsr_get_stats synchronises the stream)
splats, cfg = pkg.scenes.make_config(name)
W, H, order = cfg["width"], cfg["height"], cfg["sh_order"]
cam0 = pkg.scenes.config_camera(name, pkg.camera, W, H, order, 0)
cams = [E.camera_struct(pkg.scenes.config_camera(name, pkg.camera, W, H, order, i)) for i in range(nfr)]
band = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
# occluder = bench.py's sphere (its front pokes out of the C4 cloud), hidden = one wholly under the cloud's surface
d = np.ones((H, W), np.float32) if kind in ("far", "plain") else (pkg.scenes.sphere_occluder_depth(cam0, 3.42, 0.645) if kind == "hidden" else pkg.scenes.sphere_occluder_depth(cam0, 3.0, 0.566))
dt = torch.from_numpy(d).to("cuda")
eng = E.Engine(0)
eng.set_option(E.OPT_DEBUG_FLAGS, flags)
eng.upload(splats)
prev = eng.stats()
for i in range(nfr):
    if kind == "plain":
        eng.render_struct_to_device(cams[i], band.data_ptr())
    else:
        eng.render_struct_depth_to_device(cams[i], dt.data_ptr(), band.data_ptr())
    if sync:
        torch.cuda.synchronize()
    s = eng.stats()
    tw = eng.debug_tile_work() if sync else np.zeros((1, 1, 4), np.uint32)
    opaque = int((tw[..., 3] & 1).sum())
    if sync and kind not in ("far", "plain"):
        tcov = (d.reshape(H // 1, W)[: (H // 16) * 16, : (W // 16) * 16].reshape(H // 16, 16, W // 16, 16) < 1.0)
        full = tcov.all(axis=(1, 3)); part = tcov.any(axis=(1, 3)) & ~full
        t4 = tw[: H // 16, : W // 16].astype(np.int64)
        for nm, msk in (("fully covered", full), ("partly covered", part), ("uncovered", ~(full | part))):
            if msk.any():
                w3 = t4[..., 3][msk]
                print("      %-15s tiles %5d: entries scanned/tile %7.0f  records gathered/tile %6.0f  wave-evals/tile %6.0f  opaque %5.2f  uncovered-opaque %5.2f  met geometry %5.2f  classic %5.2f  steps all/u %.1f/%.1f" %
                      (nm, msk.sum(), t4[..., 0][msk].mean(), t4[..., 1][msk].mean(), t4[..., 2][msk].mean(), (w3 & 1).mean(), ((w3 >> 15) & 1).mean(), ((w3 >> 14) & 1).mean(),
                       ((w3 & 1) & (1 - ((w3 >> 14) & 1))).mean(), ((w3 >> 16) & 0xff).mean(), ((w3 >> 24) & 0xff).mean()))
    if sync:
        if i > 0 and "hz_prev" in globals():
            w3 = tw[..., 3].astype(np.int64)
            op_all = (w3 & 1) != 0; u_all = ((w3 >> 15) & 1) != 0; dm = ((w3 >> 14) & 1) != 0
            predc = hz_prev[2] != 0; fin = np.isfinite(hz_prev[0])
            bad_c = predc & fin & ~op_all
            bad_u = ~predc & fin & ~(u_all | (op_all & ~dm))
            print("      promise check (host view): predicted classic %d, of which not opaque now %d; predicted geometry-limited %d, of which uncovered pixels not opaque %d" % (predc.sum(), bad_c.sum(), (~predc).sum(), bad_u.sum()))
            if bad_c.any() or bad_u.any():
                ys, xs = np.nonzero(bad_c | bad_u)
                for y, x in list(zip(ys, xs))[:6]:
                    print("        tile (%d, %d): work %s  hold %.4f  status_prev %d  depth under it min %.5f max %.5f" % (x, y, [int(v) for v in tw[y, x, :3]] + [hex(int(tw[y, x, 3]))], np.sqrt(hz_prev[0][y, x]), hz_prev[2][y, x], d[y * 16:(y + 1) * 16, x * 16:(x + 1) * 16].min(), d[y * 16:(y + 1) * 16, x * 16:(x + 1) * 16].max()))
        hz = eng.debug_horizons((W + 15) // 16, (H + 15) // 16)
        hz_prev = hz.copy()
        fin = np.isfinite(hz[0])
        print("      horizons: %d of %d tiles none (+inf); finite ones: median distance %.3f, p90 %.3f, max %.3f | raw: %d none, %d not classic | status (dilated) classic %d | covered-depth cells > 0: %d" %
              ((~fin).sum(), fin.size, np.sqrt(np.median(hz[0][fin])) if fin.any() else 0, np.sqrt(np.quantile(hz[0][fin], 0.9)) if fin.any() else 0, np.sqrt(hz[0][fin].max()) if fin.any() else 0,
               np.isinf(hz[1]).sum(), np.signbit(hz[1]).sum(), int((hz[2] != 0).sum()), int((hz[3] > 0).sum())))
    print(i, {k: s[k] - (prev[k] if k.startswith("frames_") else 0) for k in ("frames_culled", "frames_repaired", "frames_slab", "frames_jumped", "frames_resorted", "frames_requeued", "n_visible", "pairs_total", "clusters_kept", "cull_dilate", "cull_holdoff", "policy_bits")},
          "opaque tiles", opaque, "of", tw.shape[0] * tw.shape[1], "gathered", int(tw[..., 1].sum()))
    prev = s
eng.close()

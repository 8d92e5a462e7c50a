"""where the microseconds of the small kernels go (needs the -DGSR_KPROF variant copied over libgsplat_hip.so):
   wall-clock stamps of one workgroup at marked points of k_bin_place (0), k_bucket_scatter (1), k_radix_local (2, bucket 512).
   python tools/kprof.py C4"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
name = sys.argv[1]
splats, cfg = pkg.scenes.make_config(name)
W, H, order = cfg["width"], cfg["height"], cfg["sh_order"]
eng = pkg.Engine(0); eng.upload(splats)
lib = pkg.engine.load_library(); fn = lib.gsr_debug_kprof; fn.argtypes = [C.c_void_p]
band = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
acc = np.zeros((8, 16)); cnt = 0
for f in range(40):
    eng.render_struct_to_device(pkg.engine.camera_struct(pkg.scenes.config_camera(name, pkg.camera, W, H, order, f)), band.data_ptr())
    torch.cuda.synchronize()
    if f >= 10:
        b = np.zeros((8, 16), np.uint64); fn(b.ctypes.data)
        v = b.astype(np.float64)
        d = np.zeros((8, 16))
        for k in range(8):
            for i in range(1, 16):
                if v[k, i] > 0 and v[k, i - 1] > 0: d[k, i] = (v[k, i] - v[k, i - 1]) / 100.0
        acc += d; cnt += 1
labels = {0: ("k_bin_place", ["ranges (scan of the totals)", "n, tile", "zero the lane masks", "load splats + (A) masks", "-", "(S) group positions", "(B) place"]),
          1: ("k_bucket_scatter", ["n, zero LDS", "gather + LDS count", "reservations (global atomics)", "scatter"]),
          2: ("k_radix_local (bucket 512)", ["prefix of the bucket counts", "load keys", "LSD passes in LDS", "ties", "store"]),
          4: ("k_preprocess (the workgroup in the middle of the grid, first iteration)", ["prologue: prefix of the cluster counts", "find cluster + geoA/geoB arrive", "front + back (chain, horizons, colour, record)", "compaction + key/payload store"])}
for k, (nm, labs) in labels.items():
    print(nm)
    for i, l in enumerate(labs, start=1):
        print("   %-34s %6.2f us" % (l, acc[k, i] / max(cnt, 1)))

fb = lib.gsr_debug_kprof_blocks; fb.argtypes = [C.c_void_p]
blk = np.zeros((5, 2, 8192), np.uint32); fb(blk.ctypes.data)
for k, nm in ((4, "k_preprocess"), (1, "k_bucket_scatter"), (2, "k_radix_local"), (3, "k_bin_count"), (0, "k_bin_place")):
    dur, items = blk[k, 0] / 100.0, blk[k, 1].copy()
    if k in (0, 3): items[(eng.stats()["n_visible"] + 1023) // 1024:] = 0      # (entries of earlier, larger frames)
    live = items > 0
    if k == 4:
        idle = (items == 0) & (dur > 0)
        if idle.any(): print("k_preprocess      workgroups without an iteration (prologue only): %d, median %.1f us" % (int(idle.sum()), float(np.median(dur[idle]))))
    if not live.any(): continue
    print("%-17s last frame: %4d workgroups with items; items median %4d p90 %4d max %4d; workgroup time median %5.1f p90 %5.1f max %5.1f us (%d items)" % (
        nm, int(live.sum()), *np.quantile(items[live], [0.5, 0.9, 1.0]).astype(int), *np.quantile(dur[live], [0.5, 0.9, 1.0]), int(items[np.argmax(np.where(live, dur, 0))])))
    if k in (0, 3):   # which blocks are the slow ones?  (blocks are in depth order: index 0 = the nearest 1024 splats)
        idx = np.argsort(-np.where(live, dur, 0))[:8]
        print("      slowest workgroups (index: us): " + ", ".join("%d: %.1f" % (int(i), float(dur[i])) for i in idx) + "; sum over workgroups %.0f us" % float(dur[live].sum()))
print({k: eng.stats()[k] for k in ("frames_culled", "frames_repaired", "n_visible", "pairs_total")})

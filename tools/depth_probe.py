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
    print(i, {k: s[k] - (prev[k] if k.startswith("frames_") else 0) for k in ("frames_culled", "frames_repaired", "frames_slab", "frames_jumped", "frames_resorted", "frames_requeued", "n_visible", "pairs_total", "clusters_kept", "cull_dilate", "cull_holdoff", "policy_bits")},
          "opaque tiles", opaque, "of", tw.shape[0] * tw.shape[1], "gathered", int(tw[..., 1].sum()))
    prev = s
eng.close()

"""frames/s and blend time against the super-tile edge S, culling on: python tools/super_probe.py C4 1280 720 4 8 16
   (S is a lower bound: the library raises it until there are <= its bin limit of super-tiles)"""
import sys, time
sys.path.insert(0, '.')
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
name, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
splats, cfg = pkg.scenes.make_config(name)
order = cfg["sh_order"]
cams = [pkg.engine.camera_struct(pkg.scenes.config_camera(name, pkg.camera, W, H, order, i)) for i in range(140)]
band = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
for S in [int(x) for x in sys.argv[4:]]:
    eng = pkg.Engine(0)
    eng.set_option(pkg.engine.OPT_SUPER_TILE, S)
    eng.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, 1)
    eng.set_option(pkg.engine.OPT_TIMING_EVERY, 4)
    eng.upload(splats)
    for i in range(40): eng.render_struct_to_device(cams[i], band.data_ptr())
    torch.cuda.synchronize()
    s0 = eng.stats()
    t0 = time.perf_counter()
    for i in range(40, 140): eng.render_struct_to_device(cams[i], band.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    s1 = eng.stats()
    bl = (s1["blend_ms_total"] - s0["blend_ms_total"]) / max(1, s1["blend_launches"] - s0["blend_launches"])
    print("%s %dx%d S>=%d: %.1f us/frame = %.1f fps, blend %.1f us, pairs %d, culled %d repaired %d" % (
        name, W, H, S, dt * 1e6, 1 / dt, bl * 1e3, s1["pairs_total"], s1["frames_culled"] - s0["frames_culled"], s1["frames_repaired"] - s0["frames_repaired"]))
    eng.close()

"""frames/s of the plain render loop under the knobs bench.py sets (caller's stream, event sampling), small-frame sort on / off.
    python tools/regime_probe.py C5"""
import sys, time
sys.path.insert(0, '.')
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
name = sys.argv[1]
splats, cfg = pkg.scenes.make_config(name)
W, H, order = cfg["width"], cfg["height"], cfg["sh_order"]
cams = [pkg.engine.camera_struct(pkg.scenes.config_camera(name, pkg.camera, W, H, order, i)) for i in range(120)]
band = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
for use_torch_stream in (0,):
    for every in (8,):
        for local in (1, 0):
            eng = pkg.Engine(0)
            if use_torch_stream:
                st = torch.cuda.Stream()
                torch.cuda.set_stream(st)
                eng.set_stream(st.cuda_stream)
            eng.set_option(pkg.engine.OPT_LOCAL_SORT, local)
            eng.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, 1)
            eng.set_option(pkg.engine.OPT_TIMING_EVERY, every)
            eng.upload(splats)
            for i in range(20): eng.render_struct_to_device(cams[i], band.data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20, 120): eng.render_struct_to_device(cams[i], band.data_ptr())
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 100
            print("%s torch-stream %d timing-every %d local-sort %d: %.1f us/frame = %.1f fps" % (name, use_torch_stream, every, local, dt * 1e6, 1 / dt))
            s_ = eng.stats()
            print("   ", {k: s_[k] for k in ("frames", "frames_culled", "frames_repaired", "frames_requeued", "frames_resorted", "sorts_skipped", "n_visible", "pairs_total", "policy_bits", "cull_dilate", "cull_holdoff", "clusters_kept", "blend_ms_total", "blend_launches")})
            eng.close()

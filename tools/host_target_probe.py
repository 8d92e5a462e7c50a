"""PCIe-inclusive frame rate: gsr_render into a HOST buffer (what a caller without GL interop gets).  python tools/host_target_probe.py C4"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
name = sys.argv[1]
splats, cfg = pkg.scenes.make_config(name)
W, H, order = cfg["width"], cfg["height"], cfg["sh_order"]
eng = pkg.Engine(0); eng.upload(splats)
cams = [pkg.scenes.config_camera(name, pkg.camera, W, H, order, i) for i in range(60)]
import ctypes as C
hip = C.CDLL("libamdhip64.so")
cs = [pkg.engine.camera_struct(c) for c in cams]
L = pkg.engine.load_library()
for kind in ("pageable, a fresh array per frame", "pageable, re-used", "pinned (hipHostMalloc), re-used"):
    if kind.startswith("pinned"):
        p = C.c_void_p(); assert hip.hipHostMalloc(C.byref(p), C.c_size_t(W * H * 16), 0) == 0
        ptr = p.value
    else:
        buf = np.empty((H, W, 4), np.float32); buf[:] = 0; ptr = buf.ctypes.data
    for c in cs[:10]: L.gsr_render(eng.h, C.byref(c), C.c_void_p(ptr), 0)
    t0 = time.perf_counter()
    for c in cs[10:]:
        if kind.endswith("per frame"): buf = np.empty((H, W, 4), np.float32); ptr = buf.ctypes.data
        assert L.gsr_render(eng.h, C.byref(c), C.c_void_p(ptr), 0) == 0
    dt = (time.perf_counter() - t0) / 50
    print("%s %dx%d host target, %-36s %.3f ms per frame = %4.0f fps (%.1f MB per frame back over PCIe = %.1f GB/s incl. the render)" % (
        name, W, H, kind + ":", dt * 1e3, 1 / dt, W * H * 16 / 1e6, W * H * 16 / dt / 1e9))

import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
splats, cfg = pkg.scenes.make_config("C4")
eng = pkg.Engine(0); eng.upload(splats)
lib = pkg.engine.load_library(); fn = lib.gsr_debug_sw_profile; fn.argtypes = [C.c_void_p]
for f in range(8): eng.render(pkg.camera.make_camera(1920, 1080, sh_order=3, frame=f))
b = np.zeros(8, np.uint64); fn(b.ctypes.data)
v = b.astype(np.float64)
print("k_sum_work phases (us @2GHz):", [round((v[i+1]-v[i])/2000.0, 2) for i in range(0, 4)] if v[5]==0 else [round((v[i]-v[0])/2000.0,2) for i in (2,3,4,5)])

"""max / mean |product - oracle| of whole frames (the tests only check the 1e-3 bound): python tools/parity_probe.py C2 C4"""
import sys
sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); oracle = ge.load_oracle()
for name in sys.argv[1:]:
    splats, cfg = pkg.scenes.make_config(name)
    cam = pkg.camera.make_camera(cfg["width"], cfg["height"], sh_order=cfg["sh_order"], frame=7)
    eng = pkg.Engine(0)
    eng.upload(splats)
    img = eng.render(cam)
    ref = oracle.render(splats, cam, threads=oracle.max_threads())
    err = np.abs(img.astype(np.float64) - ref)
    print("%s %dx%d: max |err| %.3e, mean %.3e, 99.99th percentile %.3e, pixels beyond 1e-4: %d of %d" % (
        name, cfg["width"], cfg["height"], err.max(), err.mean(), np.quantile(err, 0.9999), int((err.max(axis=2) > 1e-4).sum()), err.shape[0] * err.shape[1]))
    eng.close()

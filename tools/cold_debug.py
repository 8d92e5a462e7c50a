"""debug: what does the policy do on cold frames (every frame jumps)?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
E = pkg.engine
splats, cfg = pkg.scenes.make_config("C4")
W, H = cfg["width"], cfg["height"]
eng = pkg.Engine(0)
eng.upload(splats)
band = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
def cam(i, d):
    return E.camera_struct(pkg.camera.make_camera(W, H, sh_order=3, frame=i, distance=d))
print("steady orbit")
for i in range(6):
    eng.render_struct_to_device(cam(i, 4.62), band.data_ptr()); st = eng.stats()
    print(i, {k: st[k] for k in ("frames_slab", "frames_culled", "frames_repaired", "policy_bits", "cull_holdoff", "cull_dilate", "n_visible")})
print("cold frames")
for i in range(14):
    eng.render_struct_to_device(cam(6 + 38 * i, 4.62 * (1.3 if i % 2 else 1.0)), band.data_ptr()); st = eng.stats()
    print(i, {k: st[k] for k in ("frames_slab", "frames_culled", "frames_repaired", "policy_bits", "cull_holdoff", "cull_dilate", "n_visible")})

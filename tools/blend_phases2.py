"""k_blend under back-to-back frames (what bench.py times): per-phase wave time and the launch's span, for the small-frame sort on / off.
Needs the -DBL_PROFILE variant copied over libgsplat_hip.so.   python tools/blend_phases2.py C5 <local-sort 0|1> [far|occluder: depth-tested frames]"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
name = sys.argv[1]; local = int(sys.argv[2])
dkind = sys.argv[3] if len(sys.argv) > 3 else None
splats, cfg = pkg.scenes.make_config(name)
W, H, order = cfg["width"], cfg["height"], cfg["sh_order"]
eng = pkg.Engine(0)
eng.set_option(pkg.engine.OPT_LOCAL_SORT, local)
eng.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, 1)
eng.upload(splats)
lib = pkg.engine.load_library()
fn = lib.gsr_debug_blend_profile
fn.argtypes = [C.c_void_p, C.c_int]
band = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
cams = [pkg.engine.camera_struct(pkg.scenes.config_camera(name, pkg.camera, W, H, order, i)) for i in range(40)]
dt_ = None
if dkind:
    cam0 = pkg.scenes.config_camera(name, pkg.camera, W, H, order, 0)
    d_ = np.ones((H, W), np.float32) if dkind == "far" else pkg.scenes.sphere_occluder_depth(cam0, 3.0, 0.566)
    dt_ = torch.from_numpy(d_).to("cuda")
def frame(i):
    if dt_ is None: eng.render_struct_to_device(cams[i], band.data_ptr())
    else: eng.render_struct_depth_to_device(cams[i], dt_.data_ptr(), band.data_ptr())
for i in range(39): frame(i)
torch.cuda.synchronize()
fn(None, 1)
import time
t0 = time.perf_counter()
for i in range(20, 40): frame(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
buf = np.zeros((65536, 4, 16), dtype=np.uint64)
fn(buf.ctypes.data, 0)
live = buf[:, :, 9] > 0
# (the buffer is indexed by workgroup: rows a shorter launch did not overwrite are stale -- keep the LAST launch, the rows that ended within 2 ms of the latest end)
live &= buf[:, :, 12] + np.uint64(200000) >= buf[:, :, 12][live].max()
v = buf[live].astype(np.float64)
names = ["prologue", "scan (+its barriers)", "batch bookkeeping", "gather + quadrant test + staging", "composite",
         "batch tail", "wait: batch-end barrier", "-", "epilogue (store)"]
tot = v[:, :9].sum()
start, end = v[:, 11], v[:, 12]
span = (end.max() - start.min()) / 100.0
res = (end - start).sum() / 100.0
print("%s local-sort %d: %.1f us per frame (host clock); last launch: %d waves, span %.1f us, summed residency %.0f us = %.0f waves resident on average" %
      (name, local, dt * 1e6, v.shape[0], span, res, res / span))
for n, x in zip(names, v[:, :9].sum(axis=0)):
    print("  %-34s %6.2f %%   %7.2f us/wave" % (n, 100 * x / tot, x / tot * res / v.shape[0]))
# lane occupancy of the record evaluations (round 6): lanes_contributing / lanes_evaluated, and how many evaluations touch one 4x8 half only
ol, oe = buf[live][:, 13].astype(np.float64).sum(), buf[live][:, 14].astype(np.float64).sum()
oh = (buf[live][:, 15] & np.uint64(0xffffffff)).astype(np.float64).sum(); on = (buf[live][:, 15] >> np.uint64(32)).astype(np.float64).sum()
if oe > 0:
    print("lane occupancy: %.0f record evaluations per launch (incl. the odd-list pads); lanes contributing / lanes evaluated = %.3f; evaluations that touch no lane %.3f, "
          "only one 4x8 half of the quadrant %.3f (of those that touch any: %.3f)" % (oe, ol / (64.0 * oe), on / oe, oh / oe, oh / max(oe - on, 1.0)))
life = (end - start) / 100.0
# where do the longest-lived waves (the launch's tail) spend their time?
thr = np.quantile(life, 0.99)
vv = v[life >= thr]
tt = vv[:, :9].sum()
print("the %d waves that live >= %.1f us (p99):" % (vv.shape[0], thr))
for n, x in zip(names, vv[:, :9].sum(axis=0)):
    print("  %-34s %6.2f %%   %7.2f us/wave" % (n, 100 * x / tt, x / tt * ((vv[:, 12] - vv[:, 11]).sum() / 100.0) / vv.shape[0]))
print("  rounds (batches) per wave: median %.0f  max %.0f" % (np.median(vv[:, 10]), vv[:, 10].max()))
# which waves are the tail?  by rounds, and by where in the launch they started
order = np.argsort(-life)[:16]
print("the 16 longest-lived waves: life us / start us after launch / rounds / composite share:", "  ".join("%.0f/%.0f/%d/%.2f" % (life[i], (start[i] - start.min()) / 100.0, v[i, 10], v[i, 4] / max(v[i, :9].sum(), 1.0)) for i in order))
print("wave lifetime us: median %.1f  p90 %.1f  p99 %.1f  max %.1f" % tuple(np.quantile(life, [0.5, 0.9, 0.99, 1.0])))
t0_ = start.min()
edges = np.linspace(0, span * 100.0, 11)
occ = [(np.minimum(end - t0_, edges[i + 1]) - np.maximum(start - t0_, edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) for i in range(10)]
print("resident waves per tenth of the launch:", " ".join("%.0f" % o for o in occ))

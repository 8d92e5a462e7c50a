"""Where does k_blend's time go?  Needs a library built with -DBL_PROFILE (tools/build_variants.sh prof:"-DBL_PROFILE")
copied over libgsplat_hip.so: every wave of k_blend records the shader-clock time it spends in each phase (one row per
workgroup and wave, no atomics) plus its start/end on the constant 100 MHz clock.
Usage (GPU box): cp <variant>.so houdini-gsplat-renderer_amd/libgsplat_hip.so; python tools/blend_phases.py [C4]"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
splats, cfg = pkg.scenes.make_config(name)
eng = pkg.Engine(0)
eng.upload(splats)
lib = pkg.engine.load_library()
fn = lib.gsr_debug_blend_profile
fn.argtypes = [C.c_void_p, C.c_int]
W, H = cfg["width"], cfg["height"]
for f in range(6):
    eng.render(pkg.camera.make_camera(W, H, sh_order=cfg.get("sh_order", 3), frame=f))
buf = np.zeros((65536, 4, 16), dtype=np.uint64)
fn(None, 1)
eng.render(pkg.camera.make_camera(W, H, sh_order=cfg.get("sh_order", 3), frame=6))
fn(buf.ctypes.data, 0)
live = buf[:, :, 9] > 0
v = buf[live].astype(np.float64)            # [waves, 16]
nw = v.shape[0]
names = ["prologue", "scan (+its barriers)", "batch bookkeeping", "gather + quadrant test + staging", "composite",
         "batch tail", "wait: batch-end barrier", "-", "epilogue (store)"]
tot = v[:, :9].sum()
start, end = v[:, 11], v[:, 12]
span = (end.max() - start.min()) / 100.0
res = (end - start).sum() / 100.0
print("%s: one launch, %d waves, %.2f batches per wave" % (name, nw, v[:, 10].mean()))
print("first wave start -> last wave end %.1f us; summed wave residency %.0f us = %.0f waves resident on average "
      "(6144 = 6 workgroups on every CU)" % (span, res, res / span))
print("shader clocks per us of residency: %.0f" % (tot / res))
for n, x in zip(names, v[:, :9].sum(axis=0)):
    print("  %-34s %6.2f %%   %7.2f us/wave" % (n, 100 * x / tot, x / tot * res / nw))
life = (end - start) / 100.0
print("wave lifetime us: median %.1f  p90 %.1f  p99 %.1f  max %.1f" % tuple(np.quantile(life, [0.5, 0.9, 0.99, 1.0])))
# occupancy over time: how many waves are resident in each tenth of the launch
t0 = start.min()
edges = np.linspace(0, span * 100.0, 11)
occ = [(np.minimum(end - t0, edges[i + 1]) - np.maximum(start - t0, edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) for i in range(10)]
print("resident waves per tenth of the launch:", " ".join("%.0f" % o for o in occ))

"""Where does k_blend's time go?  Needs a library built with -DBL_PROFILE (tools/build_variants.sh prof:"-DBL_PROFILE")
copied over libgsplat_hip.so: every wave of k_blend sums the shader-clock time it spends in each phase.
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
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
W, H = cfg["width"], cfg["height"]
for f in range(6):
    eng.render(pkg.camera.make_camera(W, H, sh_order=cfg.get("sh_order", 3), frame=f))
out = (C.c_ulonglong * 16)()
fn(out, 1)
frames = 10
for f in range(6, 6 + frames):
    eng.render(pkg.camera.make_camera(W, H, sh_order=cfg.get("sh_order", 3), frame=f))
fn(out, 1)
eng.render(pkg.camera.make_camera(W, H, sh_order=cfg.get("sh_order", 3), frame=6 + frames))
one = (C.c_ulonglong * 16)()
fn(one, 0)
span = (one[12] - one[11]) / 100.0      # us (100 MHz)
res = one[13] / 100.0
print("one launch: first wave start -> last wave end %.1f us; wave residency summed %.0f us = %.0f waves resident on average "
      "(6144 = 6 workgroups on every CU)" % (span, res, res / max(span, 1e-9)))
v = np.array(list(out), dtype=np.float64)
names = ["prologue", "scan (+its barriers)", "batch bookkeeping", "gather + quadrant test + staging", "composite",
         "batch tail", "wait: batch-end barrier", "-", "epilogue (store)"]
tot = v[:9].sum()
print("%s: %d frames, %.0f waves/frame, %.1f rounds/workgroup-wave" % (name, frames, v[9] / frames, v[10] / max(v[9], 1)))
for n, x in zip(names, v[:9]):
    print("  %-32s %6.2f %%   %9.0f clocks/wave" % (n, 100 * x / tot, x / max(v[9], 1)))
print("  total clocks per wave %.0f" % (tot / max(v[9], 1)))

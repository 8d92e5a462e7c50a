"""MI355X-native Gaussian-splat rasterizer behind the Houdini GSplat plugin surface.

The directory name carries a hyphen (repo naming contract), so it is loaded
under the module name ``houdini_gsplat_renderer_amd`` by ``__graft_entry__.load_package()``.

Layout:
  csrc/       HIP kernels (gfx950) + the C ABI (include/gsplat_hip.h) + the
              HDK-free GSplatRenderer host shim (include/GSplatRenderer.h)
  build.py    hipcc driver (in-tree libgsplat_hip.so)
  engine.py   ctypes binding of the C ABI and of the GSplatRenderer wrappers
  camera.py   the reference scene's viewport camera as GL-layout matrices
  scenes.py   synthetic splat clouds of BASELINE.json's configs
  multigpu.py tile-row sharding across ranks + RCCL gather + stitch
  ply.py      INRIA 3DGS .ply ingest with the example scene's activations
"""
from . import build, camera, multigpu, ply, scenes  # noqa: F401  (no GPU needed)
from .engine import Engine, GSplatPrim, GSplatRenderer, GsrError, MultiEngine, lib_path, load_library  # noqa: F401

__all__ = ["build", "camera", "multigpu", "ply", "scenes", "Engine", "GSplatPrim", "GSplatRenderer", "GsrError", "MultiEngine", "lib_path", "load_library"]

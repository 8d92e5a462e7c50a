"""hipcc driver: builds libgsplat_hip.so IN-TREE for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgsplat_hip.so")
SOURCES = ["gsr_api.hip", "gsr_multi.cpp", "GSplatRenderer.cpp", "gsplat_ingest.cpp"]
HEADERS = ["gsr_device.h", "gsr_policy.h", "k_cluster.h", "k_preprocess.h", "k_sort.h", "k_binning.h", "k_blend.h", "k_colour.h", "k_wire.h"]
# -ffp-contract=off: only explicit fmaf() fuses (the float32 op-order contract)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-result", "-ldl", "-pthread"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm); cannot build libgsplat_hip.so")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps += [os.path.join(HERE, "..", "include", f) for f in ("gsplat_hip.h", "GSplatRenderer.h", "GSplatPrim.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return LIB

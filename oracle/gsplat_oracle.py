"""ctypes front end of the CPU ORACLE (oracle/gsplat_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by bench.py's ``cpu_baseline``
leg and by ``__graft_entry__.smoke()`` -- never by the product package.

The oracle restates the reference's per-frame path (vertex shader, fragment
shader, blend state, camera-distance argsort); see the header of
``gsplat_oracle.h`` for the reference file:line map and for how it is pinned
(SwiftShader run of the reference GLSL -> tests/golden/).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsplat_oracle.so")


class gso_frame(C.Structure):
    _fields_ = [
        ("obj_view", C.c_float * 16),
        ("object", C.c_float * 16),
        ("inv_object", C.c_float * 16),
        ("view", C.c_float * 16),
        ("proj", C.c_float * 16),
        ("cam_pos", C.c_float * 3),
        ("origin", C.c_float * 3),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("sh_order", C.c_int32),
    ]


class gso_record(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("cx", "cy", "ex", "ey", "is1", "is2", "a1x", "a1y", "b1x", "b1y", "hx", "hy", "r", "g", "b", "opacity", "la", "key", "zwin")] + \
               [("visible", C.c_int32)]


RECORD_DTYPE = np.dtype([(n, np.float32) for n in
                         ("cx", "cy", "ex", "ey", "is1", "is2", "a1x", "a1y", "b1x", "b1y", "hx", "hy", "r", "g", "b", "opacity", "la", "key", "zwin")]
                        + [("visible", np.int32)])


class gso_splats(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("P", C.c_void_p), ("Cd", C.c_void_p), ("alpha", C.c_void_p), ("scale", C.c_void_p),
        ("orient", C.c_void_p), ("shx", C.c_void_p), ("shy", C.c_void_p), ("shz", C.c_void_p),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with oracle/Makefile (gcc).  Returns the .so path."""
    src = os.path.join(_HERE, "gsplat_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
             or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "gsplat_oracle.h")))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libgsplat_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.gso_half_to_float.restype = C.c_float
        L.gso_half_to_float.argtypes = [C.c_uint16]
        L.gso_float_to_half.restype = C.c_uint16
        L.gso_float_to_half.argtypes = [C.c_float]
        L.gso_expf.restype = C.c_float
        L.gso_expf.argtypes = [C.c_float]
        L.gso_exp2f.restype = C.c_float
        L.gso_exp2f.argtypes = [C.c_float]
        L.gso_log2_opacity.restype = C.c_float
        L.gso_log2_opacity.argtypes = [C.c_float]
        L.gso_closest_sqrt_power_of_2.restype = C.c_uint
        L.gso_closest_sqrt_power_of_2.argtypes = [C.c_int]
        L.gso_preprocess.argtypes = [C.POINTER(gso_splats), C.POINTER(gso_frame), C.c_void_p]
        L.gso_argsort.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.gso_argsort_from.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.gso_host_sort_from.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        L.gso_blend_serial.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.gso_blend_parallel.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.gso_render.argtypes = [C.POINTER(gso_splats), C.POINTER(gso_frame), C.c_void_p, C.c_int]
        L.gso_render_depth.argtypes = [C.POINTER(gso_splats), C.POINTER(gso_frame), C.c_void_p, C.c_void_p]
        L.gso_render_rows.argtypes = [C.POINTER(gso_splats), C.POINTER(gso_frame), C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.gso_render_wire.argtypes = [C.POINTER(gso_splats), C.POINTER(gso_frame), C.c_void_p]
        L.gso_host_sort_only.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_void_p]
        L.gso_max_threads.restype = C.c_int
        L.gso_snap_sensitivity.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.gso_edge_mask.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_void_p]
        L.gso_storage_order.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.gso_set_tie_order.argtypes = [C.c_int]
        L.gso_set_tie_order.restype = None
        _lib = L
    return _lib


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class _SplatPack:
    """Keeps contiguous copies alive for the duration of a call."""

    def __init__(self, s):
        self.P = _c(s.P, np.float32).reshape(-1, 3)
        n = self.P.shape[0]
        self.Cd = _c(s.Cd, np.uint16).reshape(n, 3)
        self.alpha = _c(s.alpha, np.float32).reshape(n)
        self.scale = _c(s.scale, np.uint16).reshape(n, 3)
        self.orient = _c(s.orient, np.uint16).reshape(n, 4)
        has_sh = getattr(s, "shx", None) is not None
        self.shx = _c(s.shx, np.uint16).reshape(n, 16) if has_sh else None
        self.shy = _c(s.shy, np.uint16).reshape(n, 16) if has_sh else None
        self.shz = _c(s.shz, np.uint16).reshape(n, 16) if has_sh else None
        st = gso_splats()
        st.n = n
        st.P = self.P.ctypes.data
        st.Cd = self.Cd.ctypes.data
        st.alpha = self.alpha.ctypes.data
        st.scale = self.scale.ctypes.data
        st.orient = self.orient.ctypes.data
        st.shx = self.shx.ctypes.data if has_sh else None
        st.shy = self.shy.ctypes.data if has_sh else None
        st.shz = self.shz.ctypes.data if has_sh else None
        self.struct = st
        self.n = n


def make_frame(cam, origin=(0.0, 0.0, 0.0)) -> gso_frame:
    """cam: any object with obj_view/object/inv_object/view/proj (16 floats, GL
    column-major), cam_pos (3), width, height, sh_order."""
    f = gso_frame()
    for name in ("obj_view", "object", "inv_object", "view", "proj"):
        v = np.asarray(getattr(cam, name), dtype=np.float32).reshape(16)
        getattr(f, name)[:] = v.tolist()
    f.cam_pos[:] = np.asarray(cam.cam_pos, dtype=np.float32).tolist()
    f.origin[:] = np.asarray(origin, dtype=np.float32).tolist()
    f.width = int(cam.width)
    f.height = int(cam.height)
    f.sh_order = int(cam.sh_order)
    return f


def preprocess(splats, cam, origin=(0, 0, 0)) -> np.ndarray:
    pk = _SplatPack(splats)
    f = make_frame(cam, origin)
    rec = np.zeros(pk.n, dtype=RECORD_DTYPE)
    rc = lib().gso_preprocess(C.byref(pk.struct), C.byref(f), rec.ctypes.data)
    assert rc == 0
    return rec


def argsort(rec: np.ndarray, order0=None) -> np.ndarray:
    """stable ascending argsort of the keys; ties stay in the order of ``order0`` (a permutation; None = index order)"""
    rec = np.ascontiguousarray(rec)
    perm = np.zeros(rec.shape[0], dtype=np.int32)
    o = None if order0 is None else np.ascontiguousarray(order0, dtype=np.int32)
    rc = lib().gso_argsort_from(rec.ctypes.data, rec.shape[0], None if o is None else o.ctypes.data, perm.ctypes.data)
    assert rc == 0
    return perm


def blend(rec, perm, width, height, threads=1) -> np.ndarray:
    rec = np.ascontiguousarray(rec)
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    out = np.zeros((height, width, 4), dtype=np.float32)
    if threads > 1:
        rc = lib().gso_blend_parallel(rec.ctypes.data, perm.ctypes.data, rec.shape[0], width, height,
                                      out.ctypes.data, threads)
    else:
        rc = lib().gso_blend_serial(rec.ctypes.data, perm.ctypes.data, rec.shape[0], width, height,
                                    out.ctypes.data)
    assert rc == 0
    return out


def render(splats, cam, origin=(0, 0, 0), threads=1) -> np.ndarray:
    """Whole frame.  Returns float32 [H, W, 4], premultiplied, row 0 = bottom."""
    pk = _SplatPack(splats)
    f = make_frame(cam, origin)
    out = np.zeros((f.height, f.width, 4), dtype=np.float32)
    rc = lib().gso_render(C.byref(pk.struct), C.byref(f), out.ctypes.data, threads)
    assert rc == 0
    return out


def render_rows(splats, cam, row_lo: int, row_hi: int, origin=(0, 0, 0), threads: int = 0) -> np.ndarray:
    """rows [row_lo, row_hi) of the frame as an array [row_hi - row_lo, W, 4] (bit-identical to those rows of render())"""
    pk = _SplatPack(splats)
    f = make_frame(cam, origin)
    out = np.zeros((f.height, f.width, 4), dtype=np.float32)
    rc = lib().gso_render_rows(C.byref(pk.struct), C.byref(f), int(row_lo), int(row_hi) - 1, out.ctypes.data,
                               int(threads) if threads else max_threads())
    assert rc == 0
    return np.ascontiguousarray(out[row_lo:row_hi])


def render_depth(splats, cam, depth, origin=(0, 0, 0)) -> np.ndarray:
    """depth: float32 [H, W] window depth of the opaque pass (row 0 = bottom) or None"""
    pk = _SplatPack(splats)
    f = make_frame(cam, origin)
    out = np.zeros((f.height, f.width, 4), dtype=np.float32)
    d = None if depth is None else np.ascontiguousarray(depth, dtype=np.float32).reshape(f.height, f.width)
    rc = lib().gso_render_depth(C.byref(pk.struct), C.byref(f), None if d is None else d.ctypes.data, out.ctypes.data)
    assert rc == 0
    return out


def render_wire(splats, cam) -> np.ndarray:
    pk = _SplatPack(splats)
    f = make_frame(cam, (0, 0, 0))
    out = np.zeros((f.height, f.width, 4), dtype=np.float32)
    rc = lib().gso_render_wire(C.byref(pk.struct), C.byref(f), out.ctypes.data)
    assert rc == 0
    return out


def edge_mask(splats, cam, origin=(0, 0, 0), delta_px: float = 0.02, eps_log2: float = 1e-3, depth=None, eps_depth: float = 1e-6) -> np.ndarray:
    """bool [H, W]: pixels where a rasteriser's coverage rule (a pixel centre within delta_px of a quad edge), the 1/255 discard
    (alpha within 2^+-eps_log2 of it) or the depth test (|zwin - depth| <= eps_depth) may legitimately decide differently from
    the oracle for SOME visible splat -- the only places a reference-GLSL image may differ by more than the 1e-3 budget"""
    rec = np.ascontiguousarray(preprocess(splats, cam, origin))
    mask = np.zeros((int(cam.height), int(cam.width)), dtype=np.uint8)
    d = None if depth is None else np.ascontiguousarray(depth, dtype=np.float32).reshape(mask.shape)
    rc = lib().gso_edge_mask(rec.ctypes.data, rec.shape[0], int(cam.width), int(cam.height), float(delta_px), float(eps_log2),
                             None if d is None else d.ctypes.data, float(eps_depth), mask.ctypes.data)
    assert rc == 0
    return mask.astype(bool)


def snap_sensitivity(splats, cam, origin=(0, 0, 0)) -> np.ndarray:
    """float32 [H, W], 1 / pixel: sum over a pixel's fragments of T * alpha * (|kq0| |a1| + |kq1| |b1|) -- how strongly the pixel
    reacts to a sub-pixel shift of the quads covering it (a GL rasteriser snaps vertices to its sub-pixel grid)"""
    rec = np.ascontiguousarray(preprocess(splats, cam, origin))
    perm = np.ascontiguousarray(argsort(rec, storage_order(splats.P)), dtype=np.int32)
    out = np.zeros((int(cam.height), int(cam.width)), dtype=np.float32)
    rc = lib().gso_snap_sensitivity(rec.ctypes.data, perm.ctypes.data, rec.shape[0], int(cam.width), int(cam.height), out.ctypes.data)
    assert rc == 0
    return out


def storage_order(P) -> np.ndarray:
    """the contract's tie order: order[j] = index of the j-th splat in Morton order of the positions (see gsplat_oracle.h)"""
    P = _c(P, np.float32).reshape(-1, 3)
    order = np.zeros(P.shape[0], dtype=np.int32)
    rc = lib().gso_storage_order(P.ctypes.data, P.shape[0], order.ctypes.data)
    assert rc == 0
    return order


def set_tie_order(upload_order: bool):
    """True: ties in upload order (the product's GSR_OPT_STORAGE_ORDER = 0); False (default): in Morton storage order"""
    lib().gso_set_tie_order(int(bool(upload_order)))


def host_sort_only(P, cam_pos, order0=None) -> np.ndarray:
    """(distance^2, tie order) ascending; order0 None = the contract's storage order"""
    P = _c(P, np.float32).reshape(-1, 3)
    perm = np.zeros(P.shape[0], dtype=np.int32)
    cp = (C.c_float * 3)(*np.asarray(cam_pos, dtype=np.float32).tolist())
    if order0 is None:
        rc = lib().gso_host_sort_only(P.ctypes.data, P.shape[0], cp, perm.ctypes.data)
    else:
        o = np.ascontiguousarray(order0, dtype=np.int32)
        rc = lib().gso_host_sort_from(P.ctypes.data, P.shape[0], cp, o.ctypes.data, perm.ctypes.data)
    assert rc == 0
    return perm


_host_lib = None


def reference_host_stage(P, cam_pos, threads: int = 0) -> np.ndarray:
    """The reference's per-camera-move CPU work (squared distances + parallel comparison argsort of the
    indices, src/GSplatRenderer.C:188-208) -- oracle/host_stage_ref.cpp; used to TIME that stage on the
    bench box.  Unstable like the reference's tbb::parallel_sort: ties come back in arbitrary order."""
    global _host_lib
    if _host_lib is None:
        path = os.path.join(_HERE, "libgsplat_hoststage.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", _HERE, "libgsplat_hoststage.so"], check=True, stdout=subprocess.DEVNULL)
        _host_lib = C.CDLL(path)
        _host_lib.gso_reference_host_stage.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_void_p, C.c_int]
    P = _c(P, np.float32).reshape(-1, 3)
    perm = np.zeros(P.shape[0], dtype=np.int32)
    cp = (C.c_float * 3)(*np.asarray(cam_pos, dtype=np.float32).tolist())
    rc = _host_lib.gso_reference_host_stage(P.ctypes.data, P.shape[0], cp, perm.ctypes.data, int(threads))
    assert rc == 0
    return perm


def max_threads() -> int:
    return int(lib().gso_max_threads())


def half_to_float(h: int) -> float:
    return float(lib().gso_half_to_float(int(h)))


def float_to_half(f: float) -> int:
    return int(lib().gso_float_to_half(float(f)))


def expf(x: float) -> float:
    return float(lib().gso_expf(float(x)))


def exp2f(x: float) -> float:
    return float(lib().gso_exp2f(float(x)))


LOG2_255 = float(np.float32(7.99435343685886))    # GSO_LOG2_255: a fragment is discarded iff la - |kappa q|^2 < -LOG2_255


def log2_opacity(x: float) -> float:
    return float(lib().gso_log2_opacity(float(x)))


def closest_sqrt_power_of_2(n: int) -> int:
    return int(lib().gso_closest_sqrt_power_of_2(int(n)))

"""ctypes binding of libgsplat_hip.so: the C ABI (include/gsplat_hip.h) and the flat wrappers
of the GSplatRenderer host shim (include/GSplatRenderer.h).

There is NO CPU fallback: if the library is missing or no GPU is visible, every
render path raises ``GsrError``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

_LIB = None


class GsrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"gsplat_hip error {code}: {msg}")
        self.code = code


class gsr_camera(C.Structure):
    _fields_ = [("obj_view", C.c_float * 16), ("object", C.c_float * 16), ("inv_object", C.c_float * 16),
                ("view", C.c_float * 16), ("proj", C.c_float * 16), ("cam_pos", C.c_float * 3),
                ("width", C.c_int32), ("height", C.c_int32), ("sh_order", C.c_int32)]


class gsr_stats(C.Structure):
    _fields_ = [("n_splats", C.c_int64), ("n_visible", C.c_int64), ("pairs_total", C.c_int64),
                ("pairs_consumed", C.c_int64), ("tiles_x", C.c_int32), ("tiles_y", C.c_int32),
                ("record_bytes", C.c_int32), ("pair_bytes", C.c_int32),
                ("ms_preprocess", C.c_float), ("ms_depth_sort", C.c_float), ("ms_emit", C.c_float),
                ("ms_tile_sort", C.c_float), ("ms_blend", C.c_float), ("ms_total", C.c_float),
                ("blend_ms_total", C.c_double), ("blend_launches", C.c_int64),
                ("blend_pairs_consumed_total", C.c_int64), ("frame_ms_total", C.c_double), ("frames", C.c_int64),
                ("entries_scanned", C.c_int64), ("blend_entries_scanned_total", C.c_int64),
                ("super_tile", C.c_int32), ("stiles_x", C.c_int32), ("stiles_y", C.c_int32), ("reserved_", C.c_int32),
                ("blend_wave_evals_total", C.c_int64), ("stage_ms_total", C.c_double * 5),
                ("stage_frames", C.c_int64), ("sorts_skipped", C.c_int64), ("frames_requeued", C.c_int64),
                ("lazy_redo_tiles", C.c_int64), ("lazy_colours_total", C.c_int64), ("frames_truncated", C.c_int64),
                ("frames_culled", C.c_int64), ("frames_repaired", C.c_int64),
                ("clusters_total", C.c_int64), ("clusters_kept", C.c_int64),
                ("policy_bits", C.c_int32), ("cull_dilate", C.c_int32), ("cull_holdoff", C.c_int32), ("reserved2_", C.c_int32),
                ("frames_resorted", C.c_int64), ("frames_slab", C.c_int64), ("frames_jumped", C.c_int64), ("frames_lazy", C.c_int64),
                ("uploads", C.c_int64), ("upload_ms", C.c_double * 6)]

    def as_dict(self) -> dict:
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["stage_ms_total"] = list(self.stage_ms_total)
        d["upload_ms"] = list(self.upload_ms)
        return d


class gsr_debug_record(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("cx", "cy", "a1x", "a1y", "b1x", "b1y", "hx", "hy", "r", "g", "b", "la", "key")] + \
               [("visible", C.c_int32)]


DEBUG_RECORD_DTYPE = np.dtype([(n, np.float32) for n in
                               ("cx", "cy", "a1x", "a1y", "b1x", "b1y", "hx", "hy", "r", "g", "b", "la", "key")]
                              + [("visible", np.int32)])


class GSplatRenderContext(C.Structure):
    _fields_ = [("obj_view", C.c_float * 16), ("object", C.c_float * 16), ("inv_object", C.c_float * 16),
                ("view", C.c_float * 16), ("proj", C.c_float * 16), ("width", C.c_int32), ("height", C.c_int32),
                ("target", C.c_void_p), ("target_is_device", C.c_int32),
                ("depth", C.c_void_p), ("depth_is_device", C.c_int32)]


class gsr_raw_attrs(C.Structure):
    _fields_ = [("P", C.c_void_p), ("Cd", C.c_void_p), ("alpha", C.c_void_p), ("scale", C.c_void_p), ("orient", C.c_void_p),
                ("sh_scheme", C.c_int32), ("sh_vec3_per_point", C.c_int32), ("sh_array", C.c_void_p), ("sh_ptr", C.POINTER(C.c_void_p))]


class gsplat_attrs(C.Structure):
    _fields_ = [("count", C.c_int64), ("P", C.c_void_p), ("Cd", C.c_void_p), ("opacity", C.c_void_p), ("Alpha", C.c_void_p),
                ("scale", C.c_void_p), ("orient", C.c_void_p), ("sh_coefficients", C.c_void_p),
                ("sh_coefficients_len", C.c_int32), ("sh", C.POINTER(C.c_void_p)), ("f_rest", C.POINTER(C.c_void_p)),
                ("sh_order", C.POINTER(C.c_int32)), ("explicit_camera_pos", C.POINTER(C.c_float))]


TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_COPY = 0, 1, 2
MAX_DIM = 16384         # GSR_MAX_DIM (include/gsplat_hip.h): largest framebuffer width / height
MISSING_CD, MISSING_OPACITY, MISSING_SCALE, MISSING_ORIENT, MISSING_SH, BAD_SH_ORDER = 1, 2, 4, 8, 16, 32
OPT_XCD_SWIZZLE, OPT_STAGE_TIMING, OPT_SORT_CACHE, OPT_SUPER_TILE, OPT_DEBUG_FLAGS, OPT_FRAMES_IN_FLIGHT, OPT_DEFERRED_CHECK, OPT_LAZY_COLOUR, OPT_SHARD_LAYOUT = 1, 2, 3, 4, 5, 6, 7, 8, 9
OPT_OCCLUSION_CULL = 10
OPT_TIMING_EVERY = 11
OPT_CLUSTER_CULL = 12
OPT_STORAGE_ORDER = 13
OPT_CULL_DILATE = 14
OPT_LOCAL_SORT = 15
OPT_FRONT_SLAB = 16

# every symbol include/gsplat_hip.h and include/GSplatRenderer.h declare
C_ABI_SYMBOLS = [
    "gsr_device_count", "gsr_create", "gsr_destroy", "gsr_last_error", "gsr_version", "gsr_set_stream",
    "gsr_upload_begin", "gsr_upload_append", "gsr_upload_append_raw", "gsr_upload_end", "gsr_upload_abort", "gsr_upload", "gsr_set_row_shard", "gsr_band_rows",
    "gsr_stitch_bands", "gsr_render", "gsr_render_depth", "gsr_render_wire", "gsr_render_wire_over", "gsr_synchronize", "gsr_get_stats", "gsr_stats_reset", "gsr_set_option",
    "gsr_debug_read_records", "gsr_debug_read_depth_order", "gsr_debug_read_storage_order", "gsr_debug_read_tile_lists", "gsr_debug_sort_pairs", "gsr_debug_sort_pairs_local", "gsr_debug_policy", "gsr_debug_policy_state",
    "gsr_debug_read_tile_work", "gsr_debug_read_horizons",
    "gsr_multi_create", "gsr_multi_destroy", "gsr_multi_count", "gsr_multi_transport", "gsr_multi_context",
    "gsr_multi_set_stream", "gsr_multi_set_option", "gsr_multi_upload_begin", "gsr_multi_upload_append",
    "gsr_multi_upload_end", "gsr_multi_upload_abort", "gsr_multi_upload", "gsr_multi_render", "gsr_multi_render_depth",
    "gsr_multi_synchronize", "gsr_multi_get_stats", "gsr_multi_comm_info", "gsr_multi_gather_stats",
    "gsr_comm_available", "gsr_comm_get_unique_id", "gsr_comm_init", "gsr_comm_info", "gsr_comm_destroy", "gsr_comm_render",
    "gsplat_renderer_create_multi", "gsplat_renderer_multi",
    "gsplat_prim_create", "gsplat_prim_destroy", "gsplat_prim_update", "gsplat_prim_render", "gsplat_prim_missing",
    "gsplat_prim_sh_order", "gsplat_prim_has_sh", "gsplat_prim_array",
    "gsplat_renderer_create", "gsplat_renderer_get_instance", "gsplat_renderer_destroy",
    "gsplat_renderer_register_update", "gsplat_renderer_include_in_render_pass",
    "gsplat_renderer_flush_entries_for_matching_detail", "gsplat_renderer_generate_render_geometry",
    "gsplat_renderer_render", "gsplat_renderer_post_render", "gsplat_renderer_redraw", "gsplat_renderer_set_rendering_enabled",
    "gsplat_renderer_set_explicit_camera_pos", "gsplat_renderer_set_spherical_harmonics_order",
    "gsplat_renderer_query", "gsplat_renderer_get_origin", "gsplat_renderer_get_last_camera_pos",
    "gsplat_renderer_engine", "gsplat_closest_sqrt_power_of_2", "gsplat_quantize_half",
    "gsplat_pack_sh_from_vec3", "gsplat_pack_sh_from_frest", "gsplat_pack_sh_from_array",
]


POLICY_FIELDS = ("cull_pays", "cull_weak", "vis_unculled", "cull_holdoff", "cull_backoff", "cull_streak", "cull_dilate", "opt_dilate",
                 "slab_holdoff", "local_fails", "local_holdoff")


def lib_path() -> str:
    """the in-tree library; GSR_LIBRARY names another build of the same sources (the sanitizer builds of tools/build_sanitized.sh)"""
    return os.environ.get("GSR_LIBRARY") or _build.LIB


def load_library() -> C.CDLL:
    """dlopen the in-tree libgsplat_hip.so (never builds implicitly on a GPU box: the .so travels)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise GsrError(-100, f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950) first")
    L = C.CDLL(path)
    vp, i32, i64, f32p = C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_float)
    L.gsr_device_count.restype = i32
    L.gsr_last_error.restype = C.c_char_p
    L.gsr_version.restype = C.c_char_p
    L.gsr_create.argtypes = [i32, C.POINTER(vp)]
    L.gsr_destroy.argtypes = [vp]
    L.gsr_destroy.restype = None
    L.gsr_set_stream.argtypes = [vp, vp]
    L.gsr_upload_begin.argtypes = [vp, i64, i32, f32p]
    L.gsr_upload_append.argtypes = [vp, i64] + [vp] * 8
    L.gsr_upload_append_raw.argtypes = [vp, i64, C.POINTER(gsr_raw_attrs)]
    L.gsr_upload_end.argtypes = [vp]
    L.gsr_upload_abort.argtypes = [vp]
    L.gsr_upload.argtypes = [vp, i64] + [vp] * 8 + [f32p]
    L.gsr_set_row_shard.argtypes = [vp, i32, i32]
    L.gsr_band_rows.argtypes = [i32, i32, i32]
    L.gsr_stitch_bands.argtypes = [vp, vp, i32, i32, i32, vp]
    L.gsr_render.argtypes = [vp, C.POINTER(gsr_camera), vp, i32]
    L.gsr_render_depth.argtypes = [vp, C.POINTER(gsr_camera), vp, i32, vp, i32]
    L.gsr_render_wire.argtypes = [vp, C.POINTER(gsr_camera), vp, i32]
    L.gsr_render_wire_over.argtypes = [vp, C.POINTER(gsr_camera), vp, i32]
    L.gsr_synchronize.argtypes = [vp]
    L.gsr_get_stats.argtypes = [vp, C.POINTER(gsr_stats)]
    L.gsr_stats_reset.argtypes = [vp]
    L.gsr_set_option.argtypes = [vp, i32, i32]
    L.gsr_debug_read_records.argtypes = [vp, vp, i64]
    L.gsr_debug_read_depth_order.argtypes = [vp, vp, i64, C.POINTER(C.c_int64)]
    L.gsr_debug_read_tile_lists.argtypes = [vp, vp, vp, i64, vp, i64]
    L.gsr_debug_read_storage_order.argtypes = [vp, vp, i64]
    L.gsr_debug_sort_pairs.argtypes = [vp, vp, vp, i64, i32]
    L.gsr_debug_sort_pairs_local.argtypes = [vp, vp, vp, i64, i32, C.c_uint32, i32]
    L.gsr_debug_policy.argtypes = [vp, i32, C.c_longlong, C.c_longlong]
    L.gsr_debug_policy_state.argtypes = [vp, vp]
    L.gsr_debug_read_tile_work.argtypes = [vp, vp, i64]
    L.gsr_debug_read_horizons.argtypes = [vp, vp, i64]
    # host shim wrappers
    L.gsplat_renderer_create.restype = vp
    L.gsplat_renderer_create.argtypes = [i32]
    L.gsplat_renderer_get_instance.restype = vp
    L.gsplat_renderer_destroy.argtypes = [vp]
    L.gsplat_renderer_destroy.restype = None
    L.gsplat_renderer_register_update.argtypes = [vp, C.c_uint64, C.POINTER(C.c_int64), i64, i64, f32p] + [vp] * 8 + \
                                                 [i64, C.c_char_p, i32]
    L.gsplat_renderer_include_in_render_pass.argtypes = [vp, C.c_char_p]
    L.gsplat_renderer_include_in_render_pass.restype = None
    L.gsplat_renderer_flush_entries_for_matching_detail.argtypes = [vp, C.c_char_p]
    L.gsplat_renderer_flush_entries_for_matching_detail.restype = None
    L.gsplat_renderer_generate_render_geometry.argtypes = [vp, C.POINTER(GSplatRenderContext)]
    L.gsplat_renderer_generate_render_geometry.restype = None
    L.gsplat_renderer_render.argtypes = [vp, C.POINTER(GSplatRenderContext), i32]
    L.gsplat_renderer_render.restype = None
    L.gsplat_renderer_post_render.argtypes = [vp]
    L.gsplat_renderer_post_render.restype = None
    L.gsplat_renderer_redraw.argtypes = [vp, C.POINTER(C.c_char_p), i32, C.POINTER(GSplatRenderContext), i32]
    L.gsplat_renderer_redraw.restype = None
    L.gsplat_renderer_set_rendering_enabled.argtypes = [vp, i32]
    L.gsplat_renderer_set_rendering_enabled.restype = None
    L.gsplat_renderer_set_explicit_camera_pos.argtypes = [vp, f32p]
    L.gsplat_renderer_set_explicit_camera_pos.restype = None
    L.gsplat_renderer_set_spherical_harmonics_order.argtypes = [vp, i32]
    L.gsplat_renderer_set_spherical_harmonics_order.restype = None
    L.gsplat_renderer_query.argtypes = [vp, i32, C.c_char_p]
    L.gsplat_renderer_query.restype = i64
    L.gsplat_renderer_get_origin.argtypes = [vp, f32p]
    L.gsplat_renderer_get_origin.restype = None
    L.gsplat_renderer_get_last_camera_pos.argtypes = [vp, f32p]
    L.gsplat_renderer_get_last_camera_pos.restype = None
    L.gsplat_renderer_engine.argtypes = [vp]
    L.gsplat_renderer_engine.restype = vp
    L.gsplat_closest_sqrt_power_of_2.argtypes = [i32]
    L.gsplat_closest_sqrt_power_of_2.restype = C.c_uint
    L.gsplat_quantize_half.argtypes = [vp, vp, i64]
    L.gsplat_quantize_half.restype = None
    L.gsplat_pack_sh_from_vec3.argtypes = [C.POINTER(vp), i64, vp, vp, vp]
    L.gsplat_pack_sh_from_vec3.restype = None
    L.gsplat_pack_sh_from_frest.argtypes = [C.POINTER(vp), i64, vp, vp, vp]
    L.gsplat_pack_sh_from_frest.restype = None
    L.gsplat_pack_sh_from_array.argtypes = [vp, i64, i32, vp, vp, vp]
    L.gsplat_pack_sh_from_array.restype = None
    # several GPUs
    L.gsr_multi_create.argtypes = [C.POINTER(C.c_int), i32, i32, C.POINTER(vp)]
    L.gsr_multi_destroy.argtypes = [vp]
    L.gsr_multi_destroy.restype = None
    L.gsr_multi_count.argtypes = [vp]
    L.gsr_multi_transport.argtypes = [vp]
    L.gsr_multi_context.argtypes = [vp, i32]
    L.gsr_multi_context.restype = vp
    L.gsr_multi_set_stream.argtypes = [vp, vp]
    L.gsr_multi_set_option.argtypes = [vp, i32, i32]
    L.gsr_multi_upload_begin.argtypes = [vp, i64, i32, f32p]
    L.gsr_multi_upload_append.argtypes = [vp, i64] + [vp] * 8
    L.gsr_multi_upload_end.argtypes = [vp]
    L.gsr_multi_upload_abort.argtypes = [vp]
    L.gsr_multi_upload.argtypes = [vp, i64] + [vp] * 8 + [f32p]
    L.gsr_multi_render.argtypes = [vp, C.POINTER(gsr_camera), vp, i32]
    L.gsr_multi_render_depth.argtypes = [vp, C.POINTER(gsr_camera), vp, i32, vp, i32]
    L.gsr_multi_synchronize.argtypes = [vp]
    L.gsr_multi_get_stats.argtypes = [vp, i32, C.POINTER(gsr_stats)]
    L.gsr_multi_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gsr_multi_gather_stats.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.gsr_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gsr_comm_get_unique_id.argtypes = [vp]
    L.gsr_comm_init.argtypes = [vp, vp, i32, i32]
    L.gsr_comm_destroy.argtypes = [vp]
    L.gsr_comm_render.argtypes = [vp, C.POINTER(gsr_camera), vp, i32, vp]
    L.gsplat_renderer_create_multi.argtypes = [C.POINTER(C.c_int), i32, i32]
    L.gsplat_renderer_create_multi.restype = vp
    L.gsplat_renderer_multi.argtypes = [vp]
    L.gsplat_renderer_multi.restype = vp
    # the viewport primitive's part (N1 ingest)
    L.gsplat_prim_create.argtypes = [vp]
    L.gsplat_prim_create.restype = vp
    L.gsplat_prim_destroy.argtypes = [vp]
    L.gsplat_prim_destroy.restype = None
    L.gsplat_prim_update.argtypes = [vp, C.c_uint64, C.POINTER(C.c_int64), i64, C.POINTER(gsplat_attrs), f32p, C.c_char_p, i32]
    L.gsplat_prim_render.argtypes = [vp, i32]
    L.gsplat_prim_render.restype = None
    L.gsplat_prim_missing.argtypes = [vp]
    L.gsplat_prim_missing.restype = C.c_uint
    L.gsplat_prim_sh_order.argtypes = [vp]
    L.gsplat_prim_has_sh.argtypes = [vp]
    L.gsplat_prim_array.argtypes = [vp, i32]
    L.gsplat_prim_array.restype = vp
    _LIB = L
    return L


def _check(rc: int):
    if rc != 0:
        raise GsrError(rc, load_library().gsr_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return None if a is None else a.ctypes.data


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def camera_struct(cam) -> gsr_camera:
    s = gsr_camera()
    for name in ("obj_view", "object", "inv_object", "view", "proj"):
        getattr(s, name)[:] = np.asarray(getattr(cam, name), dtype=np.float32).reshape(16).tolist()
    s.cam_pos[:] = np.asarray(cam.cam_pos, dtype=np.float32).tolist()
    s.width, s.height, s.sh_order = int(cam.width), int(cam.height), int(cam.sh_order)
    return s


class _Arrays:
    """contiguous, correctly typed views of a Splats-like object (kept alive during the call)"""

    def __init__(self, s):
        self.P = np.ascontiguousarray(s.P, dtype=np.float32).reshape(-1, 3)
        n = self.n = self.P.shape[0]
        self.Cd = np.ascontiguousarray(s.Cd, dtype=np.uint16).reshape(n, 3)
        self.alpha = np.ascontiguousarray(s.alpha, dtype=np.float32).reshape(n)
        self.scale = np.ascontiguousarray(s.scale, dtype=np.uint16).reshape(n, 3)
        self.orient = np.ascontiguousarray(s.orient, dtype=np.uint16).reshape(n, 4)
        sh = getattr(s, "shx", None) is not None
        self.shx = np.ascontiguousarray(s.shx, dtype=np.uint16).reshape(n, 16) if sh else None
        self.shy = np.ascontiguousarray(s.shy, dtype=np.uint16).reshape(n, 16) if sh else None
        self.shz = np.ascontiguousarray(s.shz, dtype=np.uint16).reshape(n, 16) if sh else None

    def ptrs(self):
        return [_ptr(a) for a in (self.P, self.Cd, self.alpha, self.scale, self.orient, self.shx, self.shy, self.shz)]


class Engine:
    """One libgsplat_hip context = one GPU.  Thin, 1:1 over the gsr_* C ABI."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        h = C.c_void_p()
        _check(self.L.gsr_create(int(device), C.byref(h)))
        self.h = h
        self.device = device
        self.shard = (0, 1)

    def close(self):
        if getattr(self, "h", None):
            self.L.gsr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- staging
    def upload(self, splats, origin=(0.0, 0.0, 0.0)):
        a = _Arrays(splats)
        _check(self.L.gsr_upload(self.h, a.n, *a.ptrs(), _f3(origin)))
        return a.n

    def upload_parts(self, parts, origin=(0.0, 0.0, 0.0)):
        arrs = [_Arrays(p) for p in parts]
        has_sh = bool(arrs) and all(a.shx is not None for a in arrs)
        _check(self.L.gsr_upload_begin(self.h, sum(a.n for a in arrs), int(has_sh), _f3(origin)))
        for a in arrs:
            p = a.ptrs()
            if not has_sh:
                p[5:] = [None, None, None]
            _check(self.L.gsr_upload_append(self.h, a.n, *p))
        _check(self.L.gsr_upload_end(self.h))

    def upload_raw(self, attrs: dict, origin=(0.0, 0.0, 0.0)):
        """raw float32 point attributes (dict keyed like a Houdini detail: P, Cd, opacity / Alpha resolved by the caller into
        'alpha', scale, orient, and ONE of sh_coefficients / sh1..sh15 / f_rest_0..44): quantised and packed on the GPU"""
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        keep = {k: f32(v) for k, v in attrs.items()}
        n = int(keep["P"].reshape(-1, 3).shape[0])
        a = gsr_raw_attrs()
        for name in ("P", "Cd", "alpha", "scale", "orient"):
            setattr(a, name, keep[name].ctypes.data if name in keep else None)
        ptrs = None
        if "sh_coefficients" in keep:
            a.sh_scheme, a.sh_array = 1, keep["sh_coefficients"].ctypes.data
            a.sh_vec3_per_point = int(keep["sh_coefficients"].size // max(n, 1) // 3)
        elif "sh1" in keep:
            ptrs = (C.c_void_p * 15)(*[keep[f"sh{k + 1}"].ctypes.data if f"sh{k + 1}" in keep else None for k in range(15)])
            a.sh_scheme, a.sh_ptr = 2, ptrs
        elif "f_rest_0" in keep:
            ptrs = (C.c_void_p * 45)(*[keep[f"f_rest_{k}"].ctypes.data if f"f_rest_{k}" in keep else None for k in range(45)])
            a.sh_scheme, a.sh_ptr = 3, ptrs
        _check(self.L.gsr_upload_begin(self.h, n, int(a.sh_scheme != 0), _f3(origin)))
        try:
            _check(self.L.gsr_upload_append_raw(self.h, n, C.byref(a)))
            _check(self.L.gsr_upload_end(self.h))
        except GsrError:
            self.L.gsr_upload_abort(self.h)
            raise
        return n

    # ---- configuration
    def set_stream(self, hip_stream: int | None):
        _check(self.L.gsr_set_stream(self.h, C.c_void_p(hip_stream or 0)))

    def set_option(self, option: int, value: int):
        _check(self.L.gsr_set_option(self.h, option, value))

    def set_row_shard(self, index: int, count: int):
        _check(self.L.gsr_set_row_shard(self.h, index, count))
        self.shard = (index, count)

    def band_rows(self, height: int) -> int:
        return height if self.shard[1] == 1 else int(self.L.gsr_band_rows(height, self.shard[0], self.shard[1]))

    # ---- per frame
    def render(self, cam) -> np.ndarray:
        """synchronous render to a host array [rows, W, 4] float32 (row 0 = bottom)"""
        rows = self.band_rows(cam.height)
        out = np.empty((rows, cam.width, 4), dtype=np.float32)
        cs = camera_struct(cam)
        _check(self.L.gsr_render(self.h, C.byref(cs), out.ctypes.data, 0))
        return out

    def render_depth(self, cam, depth: np.ndarray) -> np.ndarray:
        """depth-tested frame; depth = float32 [H, W] window depth of the opaque pass (row 0 = bottom)"""
        rows = self.band_rows(cam.height)
        out = np.empty((rows, cam.width, 4), dtype=np.float32)
        d = np.ascontiguousarray(depth, dtype=np.float32).reshape(cam.height, cam.width)
        cs = camera_struct(cam)
        _check(self.L.gsr_render_depth(self.h, C.byref(cs), d.ctypes.data, 0, out.ctypes.data, 0))
        return out

    def policy_state(self) -> dict:
        """the live state of the host-side policies (csrc/gsr_policy.h)"""
        st = np.zeros(16, np.int32)
        _check(self.L.gsr_debug_policy_state(self.h, st.ctypes.data))
        return dict(zip(POLICY_FIELDS, (int(x) for x in st[:11])))

    def render_wire(self, cam) -> np.ndarray:
        """wireframe overlay (outlines of the +-2 quads, colour Cd), float32 [H, W, 4]"""
        out = np.empty((cam.height, cam.width, 4), dtype=np.float32)
        cs = camera_struct(cam)
        _check(self.L.gsr_render_wire(self.h, C.byref(cs), out.ctypes.data, 0))
        return out

    def render_wire_over(self, cam, frame: np.ndarray) -> np.ndarray:
        """wire-over display: the outlines on top of `frame` (a finished beauty frame [H, W, 4]); returns the combined image"""
        out = np.ascontiguousarray(frame, dtype=np.float32).reshape(cam.height, cam.width, 4).copy()
        cs = camera_struct(cam)
        _check(self.L.gsr_render_wire_over(self.h, C.byref(cs), out.ctypes.data, 0))
        return out

    def render_wire_over_device(self, cam, device_ptr: int):
        cs = camera_struct(cam)
        _check(self.L.gsr_render_wire_over(self.h, C.byref(cs), C.c_void_p(device_ptr), 1))

    def render_to_device(self, cam, device_ptr: int):
        cs = camera_struct(cam)
        _check(self.L.gsr_render(self.h, C.byref(cs), C.c_void_p(device_ptr), 1))

    def render_struct_to_device(self, cam_struct: gsr_camera, device_ptr: int):
        _check(self.L.gsr_render(self.h, C.byref(cam_struct), C.c_void_p(device_ptr), 1))

    def render_struct_to_host(self, cam_struct: gsr_camera, host_ptr: int):
        """gsr_render into a HOST buffer of height x width x 4 floats (what a caller without GL interop hands over)"""
        _check(self.L.gsr_render(self.h, C.byref(cam_struct), C.c_void_p(host_ptr), 0))

    def render_struct_depth_to_device(self, cam_struct: gsr_camera, depth_device_ptr: int, device_ptr: int):
        """depth-tested frame, both buffers in device memory: what the viewport hook issues on every redraw (hdk/DM_GSplatHook_hip.C)"""
        _check(self.L.gsr_render_depth(self.h, C.byref(cam_struct), C.c_void_p(depth_device_ptr), 1, C.c_void_p(device_ptr), 1))

    def stitch_bands(self, gathered_ptr: int, count: int, width: int, height: int, out_ptr: int):
        _check(self.L.gsr_stitch_bands(self.h, C.c_void_p(gathered_ptr), count, width, height, C.c_void_p(out_ptr)))

    # ---- one process per GPU: the frame's gather inside the library (gsr_comm_*)
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        _check(self.L.gsr_comm_get_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        _check(self.L.gsr_comm_init(self.h, C.c_char_p(unique_id), int(rank), int(world)))
        self.shard = (rank, world)

    def comm_render(self, cam_struct: gsr_camera, out_ptr: int, depth_ptr: int = 0):
        _check(self.L.gsr_comm_render(self.h, C.byref(cam_struct), C.c_void_p(depth_ptr or None), 1, C.c_void_p(out_ptr or None)))

    def comm_info(self):
        """(rank, ranks) as RCCL itself reports them for this context's communicator"""
        r, n = C.c_int(-1), C.c_int(0)
        _check(self.L.gsr_comm_info(self.h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def comm_destroy(self):
        _check(self.L.gsr_comm_destroy(self.h))
        self.shard = (0, 1)

    def synchronize(self):
        _check(self.L.gsr_synchronize(self.h))

    def stats(self) -> dict:
        st = gsr_stats()
        _check(self.L.gsr_get_stats(self.h, C.byref(st)))
        return st.as_dict()

    def stats_reset(self):
        _check(self.L.gsr_stats_reset(self.h))

    # ---- debug access
    def debug_records(self, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=DEBUG_RECORD_DTYPE)
        _check(self.L.gsr_debug_read_records(self.h, out.ctypes.data, n))
        return out

    def debug_depth_order(self, n: int) -> np.ndarray:
        """indices of the splats that survived culling, nearest first"""
        out = np.zeros(n, dtype=np.int32)
        cnt = C.c_int64()
        _check(self.L.gsr_debug_read_depth_order(self.h, out.ctypes.data, n, C.byref(cnt)))
        return out[:cnt.value]

    def debug_storage_order(self, n: int) -> np.ndarray:
        """perm[j] = upload index of the splat in storage slot j (Morton order of the positions by default): what breaks ties in
        the depth sort"""
        out = np.zeros(n, dtype=np.int32)
        _check(self.L.gsr_debug_read_storage_order(self.h, out.ctypes.data, n))
        return out

    def debug_tile_lists(self):
        """per-SUPER-tile [start, end) + the depth-ordered splat list of the last frame"""
        st = self.stats()
        nt = st["stiles_x"] * st["stiles_y"]
        npairs = st["pairs_total"]
        ts = np.zeros(nt, np.int32)
        te = np.zeros(nt, np.int32)
        pv = np.zeros(max(npairs, 1), np.int32)
        _check(self.L.gsr_debug_read_tile_lists(self.h, ts.ctypes.data, te.ctypes.data, nt, pv.ctypes.data, npairs))
        return ts, te, pv[:npairs]

    def debug_tile_work(self) -> np.ndarray:
        """[tiles_y, tiles_x, 4] uint32: list entries scanned / records gathered / wave-record evaluations / saturated flag
        per tile (last frame; the fourth word: bit 0 = went opaque, the rest = the 1024-entry scan step that held its first hit)"""
        st = self.stats()
        nt = st["tiles_x"] * st["tiles_y"]
        out = np.zeros((nt, 4), np.uint32)
        _check(self.L.gsr_debug_read_tile_work(self.h, out.ctypes.data, nt))
        return out.reshape(st["tiles_y"], st["tiles_x"], 4)

    def debug_horizons(self, tiles_x: int, tiles_y: int) -> np.ndarray:
        """[4, tiles_y, tiles_x] float32: dilated horizons / raw horizons (sign = status) / dilated status / covered depths (whole image)"""
        out = np.zeros((4, tiles_y, tiles_x), np.float32)
        _check(self.L.gsr_debug_read_horizons(self.h, out.ctypes.data, tiles_x * tiles_y))
        return out

    def debug_sort_pairs(self, keys: np.ndarray, vals: np.ndarray, key_bits: int = 32, local=None):
        """the pipeline's stable radix sort on host arrays; local = (bucket_lo, bucket_shift): the small-frame form (BK_BUCKETS = 1024 buckets of
        width 2^shift from lo globally, then every bucket on its own)"""
        k = np.ascontiguousarray(keys, dtype=np.uint32).copy()
        v = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        if local is None:
            _check(self.L.gsr_debug_sort_pairs(self.h, k.ctypes.data, v.ctypes.data, k.shape[0], key_bits))
        else:
            _check(self.L.gsr_debug_sort_pairs_local(self.h, k.ctypes.data, v.ctypes.data, k.shape[0], key_bits, int(local[0]), int(local[1])))
        return k, v


class GSplatRenderer:
    """Python face of the C++ GSplatRenderer host shim -- the reference's nine verbs
    (include/GSplatRenderer.h:34-56 of the reference).  ``device=-1`` gives a dry
    instance (registry/staging logic only, no GPU)."""

    Q_REGISTRY_SIZE, Q_ACTIVE_STAGED, Q_SPLAT_COUNT, Q_CAN_RENDER, Q_STAGING_COUNT, Q_RENDER_COUNT, Q_SH_PRESENT, \
        Q_LAST_STATUS, Q_ENTRY_AGE, Q_ENTRY_AGE_SINCE_ACTIVE = range(10)

    def __init__(self, device=0, transport: int = TRANSPORT_AUTO):
        """device: an int (one GPU; -1 = dry) or a sequence of device ordinals (tile rows sharded over them)"""
        self.L = load_library()
        if isinstance(device, (list, tuple)):
            arr = (C.c_int * len(device))(*[int(d) for d in device])
            self.h = self.L.gsplat_renderer_create_multi(arr, len(device), int(transport))
        else:
            self.h = self.L.gsplat_renderer_create(int(device))
        if not self.h:
            raise GsrError(-3, self.L.gsr_last_error().decode("utf-8", "replace") or "gsplat_renderer_create failed")
        self._keep = {}

    def close(self):
        if getattr(self, "h", None):
            self.L.gsplat_renderer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def registerUpdate(self, gdp: int, gversion, gVtxOffset: int, splats, splatOrigin=None) -> str:
        a = _Arrays(splats)
        origin = splats.barycenter() if splatOrigin is None else splatOrigin
        ver = (C.c_int64 * 4)(*[int(x) for x in gversion])
        buf = C.create_string_buffer(256)
        p = a.ptrs()
        n = self.L.gsplat_renderer_register_update(self.h, int(gdp), ver, int(gVtxOffset), a.n, _f3(origin), *p,
                                                   a.n if a.shx is not None else 0, buf, 256)
        if n < 0:
            raise GsrError(n, "registerUpdate failed")
        rid = buf.value.decode()
        self._keep[rid] = a  # the shim BORROWS the arrays, as the reference does
        return rid

    def includeInRenderPass(self, rid: str):
        self.L.gsplat_renderer_include_in_render_pass(self.h, rid.encode())

    def flushEntriesForMatchingDetail(self, rid: str):
        self.L.gsplat_renderer_flush_entries_for_matching_detail(self.h, rid.encode())

    @staticmethod
    def context(cam, target=None, target_is_device=False) -> GSplatRenderContext:
        r = GSplatRenderContext()
        for name in ("obj_view", "object", "inv_object", "view", "proj"):
            getattr(r, name)[:] = np.asarray(getattr(cam, name), dtype=np.float32).reshape(16).tolist()
        r.width, r.height = int(cam.width), int(cam.height)
        r.target = target
        r.target_is_device = int(bool(target_is_device))
        return r

    def generateRenderGeometry(self, r: GSplatRenderContext):
        self.L.gsplat_renderer_generate_render_geometry(self.h, C.byref(r))

    def render(self, r: GSplatRenderContext, isObjectLevel: bool = False):
        self.L.gsplat_renderer_render(self.h, C.byref(r), int(isObjectLevel))

    def postRender(self):
        self.L.gsplat_renderer_post_render(self.h)

    def redraw(self, rids, r: GSplatRenderContext, isObjectLevel: bool = False):
        """one redraw as the reference drives it (includeInRenderPass x N -> generateRenderGeometry -> render -> postRender) in ONE foreign call"""
        key = tuple(rids)
        if getattr(self, "_ids_key", None) != key:
            self._ids_key = key
            self._ids_arr = (C.c_char_p * len(rids))(*[x.encode() for x in rids])
        self.L.gsplat_renderer_redraw(self.h, self._ids_arr, len(rids), C.byref(r), int(isObjectLevel))

    def engine_stats(self) -> dict:
        """gsr_stats of the context behind the verbs (one GPU)"""
        st = gsr_stats()
        _check(self.L.gsr_get_stats(self.L.gsplat_renderer_engine(self.h), C.byref(st)))
        return st.as_dict()

    def setRenderingEnabled(self, enabled: bool):
        self.L.gsplat_renderer_set_rendering_enabled(self.h, int(enabled))

    def setExplicitCameraPos(self, pos):
        self.L.gsplat_renderer_set_explicit_camera_pos(self.h, _f3(pos))

    def setSphericalHarmonicsOrder(self, order: int):
        self.L.gsplat_renderer_set_spherical_harmonics_order(self.h, int(order))

    def query(self, what: int, rid: str | None = None) -> int:
        return int(self.L.gsplat_renderer_query(self.h, what, rid.encode() if rid else None))

    def origin(self) -> np.ndarray:
        o = (C.c_float * 3)()
        self.L.gsplat_renderer_get_origin(self.h, o)
        return np.array(list(o), dtype=np.float32)

    def lastCameraPos(self) -> np.ndarray:
        o = (C.c_float * 3)()
        self.L.gsplat_renderer_get_last_camera_pos(self.h, o)
        return np.array(list(o), dtype=np.float32)

    def frame(self, cam, rids, height=None) -> np.ndarray:
        """one redraw exactly as the reference drives it: N x GR_PrimGsplat::render marks entries
        active, then the scene hook runs generate -> render -> postRender (src/DM_GSplatHook.C:30-39)"""
        for rid in rids:
            self.includeInRenderPass(rid)
        out = np.zeros((cam.height, cam.width, 4), dtype=np.float32)
        r = self.context(cam, out.ctypes.data, False)
        self.generateRenderGeometry(r)
        self.render(r, False)
        self.postRender()
        return out


def quantize_half(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty(a.shape, dtype=np.uint16)
    load_library().gsplat_quantize_half(a.ctypes.data, out.ctypes.data, a.size)
    return out


class MultiEngine:
    """gsr_multi_*: several GPUs (or several contexts on one GPU, transport COPY) driven from this one thread."""

    def __init__(self, devices, transport: int = TRANSPORT_AUTO):
        self.L = load_library()
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        _check(self.L.gsr_multi_create(arr, len(devices), int(transport), C.byref(h)))
        self.h = h
        self.count = len(devices)

    def close(self):
        if getattr(self, "h", None):
            self.L.gsr_multi_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def transport(self) -> int:
        return int(self.L.gsr_multi_transport(self.h))

    def set_stream(self, hip_stream):
        _check(self.L.gsr_multi_set_stream(self.h, C.c_void_p(hip_stream or 0)))

    def set_option(self, option: int, value: int):
        _check(self.L.gsr_multi_set_option(self.h, option, value))

    def upload(self, splats, origin=(0.0, 0.0, 0.0)):
        a = _Arrays(splats)
        _check(self.L.gsr_multi_upload(self.h, a.n, *a.ptrs(), _f3(origin)))

    def render(self, cam, depth=None) -> np.ndarray:
        out = np.empty((cam.height, cam.width, 4), dtype=np.float32)
        cs = camera_struct(cam)
        if depth is None:
            _check(self.L.gsr_multi_render(self.h, C.byref(cs), out.ctypes.data, 0))
        else:
            d = np.ascontiguousarray(depth, dtype=np.float32).reshape(cam.height, cam.width)
            _check(self.L.gsr_multi_render_depth(self.h, C.byref(cs), d.ctypes.data, 0, out.ctypes.data, 0))
        return out

    def render_struct_to_device(self, cam_struct: gsr_camera, device_ptr: int, depth_ptr: int = 0):
        _check(self.L.gsr_multi_render_depth(self.h, C.byref(cam_struct), C.c_void_p(depth_ptr or None), 1,
                                             C.c_void_p(device_ptr), 1))

    def synchronize(self):
        _check(self.L.gsr_multi_synchronize(self.h))

    def comm_info(self):
        """([ncclCommUserRank of every rank's communicator], ncclCommCount); ([-1, ...], 0) with the COPY transport"""
        ranks = (C.c_int * self.count)()
        n = C.c_int(0)
        _check(self.L.gsr_multi_comm_info(self.h, ranks, C.byref(n)))
        return list(ranks), n.value

    def gather_stats(self, enable: int = -1):
        """(milliseconds, gathers) measured on the root's transfer stream so far; enable = 1 / 0 switches the measurement"""
        ms, k = C.c_double(0.0), C.c_int64(0)
        _check(self.L.gsr_multi_gather_stats(self.h, int(enable), C.byref(ms), C.byref(k)))
        return ms.value, k.value

    def stats(self, rank: int = 0) -> dict:
        st = gsr_stats()
        _check(self.L.gsr_multi_get_stats(self.h, rank, C.byref(st)))
        return st.as_dict()


class GSplatPrim:
    """Python face of the GSplatPrim mirror of GR_PrimGsplat (include/GSplatPrim.h): attribute ingest + the
    per-redraw verbs.  attrs: dict of float32 arrays keyed by Houdini attribute names
    (P, Cd, opacity, Alpha, scale, orient, sh_coefficients, sh1..sh15, f_rest_0..44) plus the detail attributes
    gsplat__sh_order (int) and gsplat__explicit_camera_pos (3 floats)."""

    def __init__(self, renderer: GSplatRenderer):
        self.L = load_library()
        self.R = renderer
        self.h = self.L.gsplat_prim_create(renderer.h)
        if not self.h:
            raise GsrError(-1, "gsplat_prim_create failed")
        self.id = ""
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.L.gsplat_prim_destroy(self.h)
            self.h = None

    def update(self, detail: int, version, vtx_offset: int, attrs: dict, barycenter=None) -> str:
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        keep = {k: f32(v) for k, v in attrs.items() if not k.startswith("gsplat__")}
        a = gsplat_attrs()
        a.count = int(keep["P"].reshape(-1, 3).shape[0]) if "P" in keep else 0
        for name in ("P", "Cd", "opacity", "Alpha", "scale", "orient"):
            setattr(a, name, keep[name].ctypes.data if name in keep else None)
        if "sh_coefficients" in keep:
            a.sh_coefficients = keep["sh_coefficients"].ctypes.data
            a.sh_coefficients_len = int(keep["sh_coefficients"].size // max(a.count, 1))
        sh_ptrs = (C.c_void_p * 15)(*[keep[f"sh{k + 1}"].ctypes.data if f"sh{k + 1}" in keep else None for k in range(15)])
        fr_ptrs = (C.c_void_p * 45)(*[keep[f"f_rest_{k}"].ctypes.data if f"f_rest_{k}" in keep else None for k in range(45)])
        a.sh = sh_ptrs
        a.f_rest = fr_ptrs
        order = C.c_int32(int(attrs["gsplat__sh_order"])) if "gsplat__sh_order" in attrs else None
        a.sh_order = C.pointer(order) if order is not None else None
        eye = _f3(attrs["gsplat__explicit_camera_pos"]) if "gsplat__explicit_camera_pos" in attrs else None
        a.explicit_camera_pos = eye
        ver = (C.c_int64 * 4)(*[int(x) for x in version])
        buf = C.create_string_buffer(256)
        bc = _f3(barycenter) if barycenter is not None else None
        n = self.L.gsplat_prim_update(self.h, int(detail), ver, int(vtx_offset), C.byref(a), bc, buf, 256)
        if n < 0:
            raise GsrError(n, "gsplat_prim_update failed")
        self._keep = (keep, sh_ptrs, fr_ptrs, order, eye)
        self.id = buf.value.decode()
        return self.id

    def render(self, beauty_mode: bool = True):
        self.L.gsplat_prim_render(self.h, int(beauty_mode))

    @property
    def missing(self) -> int:
        return int(self.L.gsplat_prim_missing(self.h))

    @property
    def sh_order(self) -> int:
        return int(self.L.gsplat_prim_sh_order(self.h))

    @property
    def has_sh(self) -> bool:
        return bool(self.L.gsplat_prim_has_sh(self.h))

    def arrays(self, n: int):
        """the quantised registerUpdate-layout arrays (copies) as a scenes.Splats-like namespace"""
        import types
        def grab(what, dtype, width):
            p = self.L.gsplat_prim_array(self.h, what)
            if not p:
                return None
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n * width * np.dtype(dtype).itemsize,)) \
                .view(dtype).reshape(n, width).copy()
        return types.SimpleNamespace(P=grab(0, np.float32, 3), Cd=grab(1, np.uint16, 3), alpha=grab(2, np.float32, 1).reshape(n),
                                     scale=grab(3, np.uint16, 3), orient=grab(4, np.uint16, 4), shx=grab(5, np.uint16, 16),
                                     shy=grab(6, np.uint16, 16), shz=grab(7, np.uint16, 16), n=n)

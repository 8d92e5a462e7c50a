"""Cameras for tests and the bench, expressed as the uniforms the reference's shader consumes.

Matrices are 16 floats in GL column-major order (m[c*4+r]) -- the byte layout of
Houdini's UT_Matrix4F.  The default camera is the perspective viewport saved in
the reference's example scene (hip/GSplatPlugin_simpleScene_v001.hip):
focal 50 / aperture 41.4214 (hFOV 45 deg), near 0.01, far 1e5, orbit distance 4.61995
about the origin, with the saved 3x3 rotation; ``frame`` orbits it 3 deg/frame about +Y.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# rows of the saved viewport rotation (Houdini row-vector convention: rows = camera axes in world space)
_HIP_ROT = np.array([[0.8379, -0.201042, 0.50745],
                     [2.3e-17, 0.929696, 0.368328],
                     [-0.545823, -0.308622, 0.778993]], dtype=np.float64)
_HIP_DIST = 4.61995
_HIP_P00 = 2.0 * 50.0 / 41.4214


@dataclass
class Camera:
    obj_view: np.ndarray
    object: np.ndarray
    inv_object: np.ndarray
    view: np.ndarray
    proj: np.ndarray
    cam_pos: np.ndarray
    width: int
    height: int
    sh_order: int = 3
    meta: dict = field(default_factory=dict)


def _gl(m: np.ndarray) -> np.ndarray:
    """4x4 (column-vector maths) -> 16 floats, GL column-major."""
    return np.ascontiguousarray(m.T, dtype=np.float32).reshape(16)


def perspective(p00: float, aspect_w_over_h: float, near: float, far: float) -> np.ndarray:
    p = np.zeros((4, 4), dtype=np.float64)
    p[0, 0] = p00
    p[1, 1] = p00 * aspect_w_over_h
    p[2, 2] = -(far + near) / (far - near)
    p[2, 3] = -2.0 * far * near / (far - near)
    p[3, 2] = -1.0
    return p


def frustum(left: float, right: float, bottom: float, top: float, near: float, far: float) -> np.ndarray:
    """glFrustum: an OFF-CENTRE perspective projection when left != -right or bottom != -top (a cropped / zoomed / tiled
    viewport): P02 = (r+l)/(r-l), P12 = (t+b)/(t-b)"""
    p = np.zeros((4, 4), dtype=np.float64)
    p[0, 0] = 2.0 * near / (right - left)
    p[1, 1] = 2.0 * near / (top - bottom)
    p[0, 2] = (right + left) / (right - left)
    p[1, 2] = (top + bottom) / (top - bottom)
    p[2, 2] = -(far + near) / (far - near)
    p[2, 3] = -2.0 * far * near / (far - near)
    p[3, 2] = -1.0
    return p


def orthographic(left: float, right: float, bottom: float, top: float, near: float, far: float) -> np.ndarray:
    """glOrtho (Houdini's Top / Front / Right views): clip w == 1 everywhere"""
    p = np.zeros((4, 4), dtype=np.float64)
    p[0, 0] = 2.0 / (right - left)
    p[1, 1] = 2.0 / (top - bottom)
    p[2, 2] = -2.0 / (far - near)
    p[0, 3] = -(right + left) / (right - left)
    p[1, 3] = -(top + bottom) / (top - bottom)
    p[2, 3] = -(far + near) / (far - near)
    p[3, 3] = 1.0
    return p


def look_from(rot_rows: np.ndarray, position: np.ndarray) -> np.ndarray:
    """world->camera matrix for a camera whose axes (x right, y up, z backwards) are the rows
    of ``rot_rows`` and which sits at ``position``."""
    c2w = np.eye(4)
    c2w[:3, :3] = rot_rows.T
    c2w[:3, 3] = position
    return np.linalg.inv(c2w)


def make_camera(width: int, height: int, sh_order: int = 3, frame: int = 0, distance: float = _HIP_DIST,
                p00: float = _HIP_P00, near: float = 0.01, far: float = 1.0e5,
                object_matrix: np.ndarray | None = None, rot_rows: np.ndarray | None = None,
                pivot=(0.0, 0.0, 0.0), step_deg: float = 3.0, proj_matrix: np.ndarray | None = None) -> Camera:
    """proj_matrix: a 4x4 projection (column-vector maths) instead of the symmetric perspective(p00, aspect, near, far)"""
    rot = _HIP_ROT if rot_rows is None else np.asarray(rot_rows, dtype=np.float64)
    ang = np.deg2rad(step_deg * frame)
    ry = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    rot = rot @ ry.T                       # orbit the whole rig about +Y
    pos = np.asarray(pivot, dtype=np.float64) + distance * rot[2]
    view = look_from(rot, pos)
    obj = np.eye(4) if object_matrix is None else np.asarray(object_matrix, dtype=np.float64)
    proj = perspective(p00, width / height, near, far) if proj_matrix is None else np.asarray(proj_matrix, dtype=np.float64)
    obj_view = view @ obj
    cam_pos = np.linalg.inv(view)[:3, 3]   # translation of V^-1 (src/GSplatRenderer.C:558-562)
    return Camera(obj_view=_gl(obj_view), object=_gl(obj), inv_object=_gl(np.linalg.inv(obj)), view=_gl(view),
                  proj=_gl(proj), cam_pos=cam_pos.astype(np.float32), width=int(width), height=int(height),
                  sh_order=int(sh_order), meta={"frame": frame, "distance": distance})


def rotated_in_place(cam: Camera, yaw_deg: float, pitch_deg: float = 0.0) -> Camera:
    """the same camera turned about its own position (the reference re-sorts only when the POSITION moves,
    src/GSplatRenderer.C:165-186): view' = R_cam * view; cam_pos is carried over bit for bit"""
    view = np.asarray(cam.view, dtype=np.float64).reshape(4, 4).T
    obj = np.asarray(cam.object, dtype=np.float64).reshape(4, 4).T
    a, b = np.deg2rad(yaw_deg), np.deg2rad(pitch_deg)
    ry = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]])
    rx = np.array([[1, 0, 0, 0], [0, np.cos(b), -np.sin(b), 0], [0, np.sin(b), np.cos(b), 0], [0, 0, 0, 1]])
    v2 = rx @ ry @ view
    return Camera(obj_view=_gl(v2 @ obj), object=cam.object.copy(), inv_object=cam.inv_object.copy(), view=_gl(v2), proj=cam.proj.copy(),
                  cam_pos=cam.cam_pos.copy(), width=cam.width, height=cam.height, sh_order=cam.sh_order, meta=dict(cam.meta))

"""Synthetic splat clouds for BASELINE.json's configs (distributions frozen per SURVEY 8d).

Arrays use the reference's registerUpdate() layout (include/GSplatRenderer.h:34-47 of the
reference): P f32 [n,3]; Cd/scale f16 bits [n,3]; orient f16 bits [n,4] (x,y,z,w); alpha f32 [n];
shx/shy/shz f16 bits [n,16] (coefficient j at flat index j) or None.  Quantisation is IEEE
round-to-nearest-even, which is what HDK's fpreal16 does (src/GR_GSplat.C:315-318).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

C0 = 0.2820948


@dataclass
class Splats:
    P: np.ndarray
    Cd: np.ndarray
    alpha: np.ndarray
    scale: np.ndarray
    orient: np.ndarray
    shx: np.ndarray | None = None
    shy: np.ndarray | None = None
    shz: np.ndarray | None = None

    @property
    def n(self) -> int:
        return int(self.P.shape[0])

    @property
    def has_sh(self) -> bool:
        return self.shx is not None

    def barycenter(self) -> np.ndarray:
        """GEO_PrimGsplat::baryCenter: float32 mean of the points (src/GEO_GSplat.C:338-351)."""
        if self.n == 0:
            return np.zeros(3, dtype=np.float32)
        return self.P.astype(np.float64).mean(axis=0).astype(np.float32)

    def subset(self, sl) -> "Splats":
        g = lambda a: None if a is None else np.ascontiguousarray(a[sl])
        return Splats(g(self.P), g(self.Cd), g(self.alpha), g(self.scale), g(self.orient), g(self.shx), g(self.shy), g(self.shz))


def f16bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).astype(np.float16).view(np.uint16)


def make_scene(n: int, seed: int, sh: bool = True, isotropic: bool = False, radius: float = 1.0,
               log_scale_range=(-5.5, -3.5), chunk: int = 1 << 20) -> Splats:
    """uniform-in-ball positions; scale = exp(U(lo,hi)); orient = normalised N(0,1)^4;
    opacity = sigmoid(N(0,2)); Cd = clip(0.5 + C0*N(0,1), 0, 1); f_rest ~ N(0, 0.1)."""
    rng = np.random.default_rng(seed)
    P = np.empty((n, 3), np.float32)
    Cd = np.empty((n, 3), np.uint16)
    alpha = np.empty(n, np.float32)
    scale = np.empty((n, 3), np.uint16)
    orient = np.empty((n, 4), np.uint16)
    shx = shy = shz = None
    if sh:
        shx = np.zeros((n, 16), np.uint16)
        shy = np.zeros((n, 16), np.uint16)
        shz = np.zeros((n, 16), np.uint16)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        d = rng.standard_normal((m, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        r = radius * rng.random(m) ** (1.0 / 3.0)
        P[lo:lo + m] = (d * r[:, None]).astype(np.float32)
        if isotropic:
            s = np.exp(rng.uniform(log_scale_range[0], log_scale_range[1], (m, 1))).repeat(3, axis=1)
        else:
            s = np.exp(rng.uniform(log_scale_range[0], log_scale_range[1], (m, 3)))
        scale[lo:lo + m] = f16bits(s)
        q = rng.standard_normal((m, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        orient[lo:lo + m] = f16bits(q)
        alpha[lo:lo + m] = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 2.0, m)))).astype(np.float32)
        Cd[lo:lo + m] = f16bits(np.clip(0.5 + C0 * rng.standard_normal((m, 3)), 0.0, 1.0))
        if sh:
            fr = rng.normal(0.0, 0.1, (m, 45))  # f_rest_0..44, channel-major (INRIA layout)
            shx[lo:lo + m, :15] = f16bits(fr[:, 0:15])
            shy[lo:lo + m, :15] = f16bits(fr[:, 15:30])
            shz[lo:lo + m, :15] = f16bits(fr[:, 30:45])
    return Splats(P, Cd, alpha, scale, orient, shx, shy, shz)


def _attributes(rng, n, sh, log_scale_range=(-5.5, -3.5)):
    """the attribute distributions of make_scene() for n splats (positions are the caller's)"""
    scale = f16bits(np.exp(rng.uniform(log_scale_range[0], log_scale_range[1], (n, 3))))
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    orient = f16bits(q)
    alpha = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 2.0, n)))).astype(np.float32)
    Cd = f16bits(np.clip(0.5 + C0 * rng.standard_normal((n, 3)), 0.0, 1.0))
    shx = shy = shz = None
    if sh:
        fr = rng.normal(0.0, 0.1, (n, 45))
        shx = np.zeros((n, 16), np.uint16); shy = np.zeros((n, 16), np.uint16); shz = np.zeros((n, 16), np.uint16)
        shx[:, :15] = f16bits(fr[:, 0:15]); shy[:, :15] = f16bits(fr[:, 15:30]); shz[:, :15] = f16bits(fr[:, 30:45])
    return Cd, alpha, scale, orient, shx, shy, shz


def make_terrain(n: int, seed: int, sh: bool = True, extent: float = 3.0, depth: float = 0.5) -> Splats:
    """A landscape: splats fill the ground below a rolling height field (x, z in [-extent, extent], y up), `depth` thick.
    Seen from a camera above the ground (terrain_camera) a third to a half of the frame is empty SKY and the skyline crosses
    whole tile rows -- what a real capture looks like, and what per-tile depth horizons are for (a 128-px super-tile with one
    sky tile used to lose its horizon altogether)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-extent, extent, n)
    z = rng.uniform(-extent, extent, n)
    h = 0.35 * np.sin(1.3 * x + 0.4) * np.cos(1.1 * z) + 0.18 * np.sin(2.9 * x - 1.0 * z) + 0.10 * np.cos(4.3 * z + 0.7)
    y = h - depth * rng.random(n) ** 2          # denser towards the surface
    P = np.stack([x, y, z], axis=1).astype(np.float32)
    return Splats(P, *_attributes(rng, n, sh))


def make_slab(n: int, seed: int, sh: bool = True, half=(2.0, 1.0, 0.12)) -> Splats:
    """A thin wall (2 half[0] x 2 half[1], 2 half[2] thick) about the origin: face-on it is an opaque rectangle in an empty
    frame, edge-on a thin strip -- silhouettes dominate, and they sweep across the frame as the camera orbits."""
    rng = np.random.default_rng(seed)
    P = (rng.uniform(-1.0, 1.0, (n, 3)) * np.asarray(half)).astype(np.float32)
    return Splats(P, *_attributes(rng, n, sh))


def terrain_camera(camera_mod, width: int, height: int, frame: int = 0, sh_order: int = 3, distance: float = 4.2, pitch_deg: float = 14.0,
                   step_deg: float = 3.0):
    """camera above the landscape of make_terrain, looking slightly down at the pivot, orbiting about +Y by step_deg per frame"""
    p = np.deg2rad(pitch_deg)
    back = np.array([0.0, np.sin(p), np.cos(p)])           # camera z (backwards): the camera sits at pivot + distance * back
    right = np.array([1.0, 0.0, 0.0])
    up = np.cross(back, right)
    return camera_mod.make_camera(width, height, sh_order=sh_order, frame=frame, distance=distance, rot_rows=np.stack([right, up, back]),
                                  pivot=(0.0, 0.25, 0.0), step_deg=step_deg)


# BASELINE.json configs -> (n, seed, sh, isotropic, radius, width, height, sh_order)
CONFIGS = {
    "C1": dict(n=10_000, seed=1001, sh=False, isotropic=True, radius=1.0, width=512, height=512, sh_order=0),
    "C2": dict(n=100_000, seed=1002, sh=True, isotropic=False, radius=1.0, width=1280, height=720, sh_order=3),
    "C3": dict(n=1_000_000, seed=1003, sh=True, isotropic=False, radius=1.0, width=1920, height=1080, sh_order=3),
    "C4": dict(n=6_000_000, seed=1004, sh=True, isotropic=False, radius=2.0, width=1920, height=1080, sh_order=3),
    "C5": dict(n=6_000_000, seed=1005, sh=True, isotropic=False, radius=2.0, width=3840, height=2160, sh_order=3),
    # not BASELINE configs: scenes with sky and silhouettes (informational bench legs, culling tests).  B1 = the C4 cloud seen from
    # twice as far: a ball in the middle of an empty frame
    "B1": dict(n=6_000_000, seed=1004, sh=True, isotropic=False, radius=2.0, width=1920, height=1080, sh_order=3, distance=9.0),
    "T1": dict(n=4_000_000, seed=2001, sh=True, kind="terrain", width=1920, height=1080, sh_order=3),
    "S1": dict(n=2_000_000, seed=2002, sh=True, kind="slab", width=1920, height=1080, sh_order=3),
}


def config_camera(name: str, camera_mod, width: int, height: int, sh_order: int, frame: int):
    """the orbit camera of a config's frame"""
    if CONFIGS[name].get("kind") == "terrain":
        return terrain_camera(camera_mod, width, height, frame=frame, sh_order=sh_order)
    if "distance" in CONFIGS[name]:
        return camera_mod.make_camera(width, height, sh_order=sh_order, frame=frame, distance=CONFIGS[name]["distance"])
    return camera_mod.make_camera(width, height, sh_order=sh_order, frame=frame)


def make_config(name: str, n_override: int | None = None) -> tuple[Splats, dict]:
    cfg = dict(CONFIGS[name])
    if n_override is not None:
        cfg["n"] = int(n_override)
    if cfg.get("kind") == "terrain":
        return make_terrain(cfg["n"], cfg["seed"], sh=cfg["sh"]), cfg
    if cfg.get("kind") == "slab":
        return make_slab(cfg["n"], cfg["seed"], sh=cfg["sh"]), cfg
    s = make_scene(cfg["n"], cfg["seed"], sh=cfg["sh"], isotropic=cfg["isotropic"], radius=cfg["radius"])
    if name == "C3":
        # the example scene overwrites Cd with 0.5 grey before the SOP
        s.Cd[:] = f16bits(np.full((1, 3), 0.5))
    return s, cfg


INRIA_PROPERTIES = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{k}" for k in range(45)] + \
                   ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]


def make_inria_raw(n: int, seed: int, radius: float = 1.0, log_scale_range=(-5.5, -3.5)) -> np.ndarray:
    """RAW (pre-activation) INRIA 3DGS attributes with the distributions of make_scene(): what the example scene's
    `file1` node would import (the capture itself is not shipped) -- structured float32 array with
    the PLY property names; ply.splats_from_inria() applies the scene's activations to it."""
    rng = np.random.default_rng(seed)
    v = np.zeros(n, dtype=[(nm, "<f4") for nm in INRIA_PROPERTIES])
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = d * (radius * rng.random(n) ** (1.0 / 3.0))[:, None]
    v["x"], v["y"], v["z"] = p[:, 0], p[:, 1], p[:, 2]
    for k in range(3):
        v[f"f_dc_{k}"] = rng.standard_normal(n)
        v[f"scale_{k}"] = rng.uniform(log_scale_range[0], log_scale_range[1], n)     # log scale (activation: exp)
    fr = rng.normal(0.0, 0.1, (n, 45))
    for k in range(45):
        v[f"f_rest_{k}"] = fr[:, k]
    v["opacity"] = rng.normal(0.0, 2.0, n)                                            # logit (activation: sigmoid)
    q = rng.standard_normal((n, 4)) * rng.uniform(0.5, 2.0, (n, 1))                   # un-normalised (activation: normalize)
    for k in range(4):
        v[f"rot_{k}"] = q[:, k]
    return v


def write_inria_ply(path: str, v: np.ndarray) -> None:
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % v.shape[0]
    hdr += "".join(f"property float {nm}\n" for nm in v.dtype.names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(v.tobytes())

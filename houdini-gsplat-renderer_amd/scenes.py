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


def _quat_from_normal(nrm: np.ndarray, roll: np.ndarray) -> np.ndarray:
    """unit quaternions (x, y, z, w) that turn the local z axis onto `nrm` (unit vectors), with a roll about it"""
    z = np.array([0.0, 0.0, 1.0])
    axis = np.cross(np.broadcast_to(z, nrm.shape), nrm)
    s = np.linalg.norm(axis, axis=1)
    c = nrm[:, 2]
    ang = np.arctan2(s, c)
    axis = np.where(s[:, None] > 1e-9, axis / np.maximum(s, 1e-9)[:, None], np.array([1.0, 0.0, 0.0]))
    q1 = np.concatenate([axis * np.sin(ang / 2)[:, None], np.cos(ang / 2)[:, None]], axis=1)          # z -> nrm
    q0 = np.stack([np.zeros_like(roll), np.zeros_like(roll), np.sin(roll / 2), np.cos(roll / 2)], axis=1)   # roll about z first
    x1, y1, z1, w1 = q1.T
    x0, y0, z0, w0 = q0.T
    return np.stack([w1 * x0 + x1 * w0 + y1 * z0 - z1 * y0, w1 * y0 - x1 * z0 + y1 * w0 + z1 * x0,
                     w1 * z0 + x1 * y0 - y1 * x0 + z1 * w0, w1 * w0 - x1 * x0 - y1 * y0 - z1 * z0], axis=1)


def make_capture(n: int, seed: int, sh: bool = True) -> Splats:
    """R1 -- a cloud shaped like a trained CAPTURE rather than like a ball of noise (the reference's only scene imports an INRIA
    capture, hip/GSplatPlugin_simpleScene_v001.hip; the file itself is not shipped):
      * splats lie ON nested surfaces -- an object in the middle (a lumpy ellipsoid, 40 %), the floor it stands on (15 %) and the
        room around it (a shell at radius ~2.6 the camera is INSIDE of, 15 %) -- flattened along the surface normal, oriented to it;
      * sizes are log-normal with a heavy tail: most splats are a few pixels, the room's are ten times larger, and 300 background
        splats fill a good part of the screen each;
      * 30 % are FLOATERS: near-transparent (opacity ~ 0.03-0.1) blobs anywhere in the volume, which every ray crosses dozens of
        without ever saturating;
      * colour is view-dependent: first-order SH coefficients three times the size of the higher ones.
    Nothing here is friendly to tile-horizon culling: rays end on a surface only after crossing the floaters, silhouettes of the object
    against the far room leave open tiles everywhere, the big background splats reach hundreds of tiles."""
    rng = np.random.default_rng(seed)
    n_obj, n_floor, n_room = int(0.40 * n), int(0.15 * n), int(0.15 * n)
    import os
    n_big = min(int(os.environ.get("GSR_R1_BIG", "300")), n // 20)     # (diagnostic hook: how much of a frame the screen-filling splats cost)
    n_float = n - n_obj - n_floor - n_room - n_big

    def unit(m):
        d = rng.standard_normal((m, 3))
        return d / np.linalg.norm(d, axis=1, keepdims=True)
    # the object: r(d) = 0.55 * (1 + lumps); normal ~ radial (good enough for orientation)
    d = unit(n_obj)
    r = 0.55 * (1.0 + 0.25 * np.sin(3.0 * d[:, 0] + 1.0) * np.cos(4.0 * d[:, 1]) + 0.15 * np.sin(5.0 * d[:, 2]))
    P_obj = d * r[:, None] * np.array([1.0, 1.3, 0.9])
    P_obj += 0.004 * rng.standard_normal((n_obj, 3))
    N_obj = d
    # the floor: a disk of radius 2.4 at y = -0.75 with gentle bumps
    rad = 2.4 * np.sqrt(rng.random(n_floor)); th = rng.uniform(0, 2 * np.pi, n_floor)
    P_floor = np.stack([rad * np.cos(th), -0.75 + 0.03 * np.sin(3 * rad * np.cos(th)) + 0.004 * rng.standard_normal(n_floor), rad * np.sin(th)], axis=1)
    N_floor = np.broadcast_to(np.array([0.0, 1.0, 0.0]), (n_floor, 3))
    # the room: a shell of radius 2.6 (upper part: the floor closes it below)
    d = unit(n_room); d[:, 1] = np.abs(d[:, 1]) * 1.0 - 0.28
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    P_room = d * (2.6 + 0.05 * rng.standard_normal(n_room))[:, None]
    N_room = -d
    # the screen-filling background splats: on the room, huge
    d = unit(n_big); d[:, 1] = np.abs(d[:, 1])
    P_big = d * 2.7
    N_big = -d
    # floaters: anywhere inside the room
    P_fl = unit(n_float) * (2.5 * rng.random(n_float) ** (1.0 / 3.0))[:, None]
    P_fl[:, 1] = np.abs(P_fl[:, 1]) * 0.9 - 0.7
    N_fl = unit(n_float)
    P = np.concatenate([P_obj, P_floor, P_room, P_big, P_fl]).astype(np.float32)
    N = np.concatenate([N_obj, N_floor, N_room, N_big, N_fl])
    m = P.shape[0]
    # sizes: log-normal in-plane, flattened along the normal
    mu = np.concatenate([np.full(n_obj, -5.0), np.full(n_floor, -4.6), np.full(n_room, -3.0), np.full(n_big, -0.4), np.full(n_float, -4.2)])
    sg = np.concatenate([np.full(n_obj, 0.55), np.full(n_floor, 0.5), np.full(n_room, 0.6), np.full(n_big, 0.35), np.full(n_float, 0.8)])
    ls = mu[:, None] + sg[:, None] * rng.standard_normal((m, 3))
    ls[:, 2] += np.where(np.arange(m) < m - n_float, -1.6, 0.0)                       # surface splats are flat; floaters are blobs
    scale = f16bits(np.exp(ls))
    orient = f16bits(_quat_from_normal(N, rng.uniform(0, 2 * np.pi, m)))
    logit = np.concatenate([rng.normal(2.5, 1.5, n_obj + n_floor), rng.normal(1.5, 1.5, n_room), rng.normal(2.0, 1.0, n_big), rng.normal(-3.0, 0.8, n_float)])
    alpha = (1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
    base = np.concatenate([np.tile([0.75, 0.45, 0.30], (n_obj, 1)), np.tile([0.35, 0.40, 0.30], (n_floor, 1)), np.tile([0.55, 0.60, 0.70], (n_room, 1)),
                           np.tile([0.60, 0.62, 0.68], (n_big, 1)), np.tile([0.5, 0.5, 0.5], (n_float, 1))])
    Cd = f16bits(np.clip(base + 0.12 * rng.standard_normal((m, 3)), 0.0, 1.0))
    shx = shy = shz = None
    if sh:
        fr = rng.normal(0.0, 0.08, (m, 45))
        for ch in range(3):
            fr[:, ch * 15: ch * 15 + 3] *= 3.0                                            # degree 1 dominates: view-dependent shading
        shx = np.zeros((m, 16), np.uint16); shy = np.zeros((m, 16), np.uint16); shz = np.zeros((m, 16), np.uint16)
        shx[:, :15] = f16bits(fr[:, 0:15]); shy[:, :15] = f16bits(fr[:, 15:30]); shz[:, :15] = f16bits(fr[:, 30:45])
    perm = rng.permutation(m)                 # a capture's points come in no particular order
    g = lambda a: None if a is None else np.ascontiguousarray(a[perm])
    return Splats(g(P), g(Cd), g(alpha), g(scale), g(orient), g(shx), g(shy), g(shz))


def fit_orbit(P: np.ndarray) -> dict:
    """an orbit for an arbitrary cloud (bench.py --ply): pivot = the median position, distance = 1.6 x the radius that holds 80 % of
    the points about it (robust against the far background points every capture has)"""
    P = np.asarray(P, dtype=np.float64)
    P = P[np.isfinite(P).all(axis=1)]
    if P.shape[0] == 0:
        return {"pivot": (0.0, 0.0, 0.0), "distance": 4.61995}
    pivot = np.median(P, axis=0)
    r = float(np.quantile(np.linalg.norm(P - pivot, axis=1), 0.8))
    return {"pivot": tuple(float(x) for x in pivot), "distance": max(1.6 * r, 1e-3)}


def register_ply_config(path: str, ply_mod, width: int = 1920, height: int = 1080, sh_order: int = 3, name: str = "PLY") -> str:
    """bench.py --ply PATH: an INRIA 3DGS capture (the example scene's activations, ply.py / SURVEY App. D) as a config of its own,
    with an orbit fitted to the cloud"""
    s = ply_mod.load_inria_ply(path)
    CONFIGS[name] = dict(n=s.n, seed=0, sh=s.has_sh, kind="ply", path=path, width=width, height=height, sh_order=sh_order if s.has_sh else 0, **fit_orbit(s.P))
    _PLY_CACHE[name] = s
    return name


_PLY_CACHE: dict = {}


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
    # R1 = a capture-shaped cloud (make_capture): surfaces, floaters, a heavy tail of sizes; the camera orbits INSIDE the room
    "R1": dict(n=3_000_000, seed=2003, sh=True, kind="capture", width=1920, height=1080, sh_order=3, distance=2.1, pivot=(0.0, -0.1, 0.0)),
}


def config_camera(name: str, camera_mod, width: int, height: int, sh_order: int, frame: int):
    """the orbit camera of a config's frame"""
    if CONFIGS[name].get("kind") == "terrain":
        return terrain_camera(camera_mod, width, height, frame=frame, sh_order=sh_order)
    if "distance" in CONFIGS[name]:
        return camera_mod.make_camera(width, height, sh_order=sh_order, frame=frame, distance=CONFIGS[name]["distance"],
                                      pivot=CONFIGS[name].get("pivot", (0.0, 0.0, 0.0)))
    return camera_mod.make_camera(width, height, sh_order=sh_order, frame=frame)


def make_config(name: str, n_override: int | None = None) -> tuple[Splats, dict]:
    cfg = dict(CONFIGS[name])
    if n_override is not None:
        cfg["n"] = int(n_override)
    if cfg.get("kind") == "terrain":
        return make_terrain(cfg["n"], cfg["seed"], sh=cfg["sh"]), cfg
    if cfg.get("kind") == "slab":
        return make_slab(cfg["n"], cfg["seed"], sh=cfg["sh"]), cfg
    if cfg.get("kind") == "capture":
        return make_capture(cfg["n"], cfg["seed"], sh=cfg["sh"]), cfg
    if cfg.get("kind") == "ply":
        s = _PLY_CACHE[name]
        return (s if n_override is None else s.subset(slice(0, int(n_override)))), cfg
    s = make_scene(cfg["n"], cfg["seed"], sh=cfg["sh"], isotropic=cfg["isotropic"], radius=cfg["radius"])
    if name == "C3":
        # the example scene overwrites Cd with 0.5 grey before the SOP
        s.Cd[:] = f16bits(np.full((1, 3), 0.5))
    return s, cfg


def sphere_occluder_depth(cam, centre_distance: float, radius: float) -> np.ndarray:
    """The opaque pass's depth buffer (float32 [H, W] window depth in [0, 1], row 0 = bottom, cleared to the far plane = 1.0) with one
    opaque SPHERE in it, centred on the view axis `centre_distance` in front of the camera: what the Houdini hook hands the renderer
    when opaque geometry sits among the splats (the reference draws with the depth test on, src/GSplatRenderer.C:595-610).  Ray-traced
    on the host in float64 through cam.proj (any perspective projection whose clip w = -z)."""
    W, H = cam.width, cam.height
    P = np.asarray(cam.proj, dtype=np.float64).reshape(4, 4).T
    i = (np.arange(W) + 0.5) / W * 2.0 - 1.0
    j = (np.arange(H) + 0.5) / H * 2.0 - 1.0
    dx = ((i - P[0, 2]) / P[0, 0])[None, :]
    dy = ((j - P[1, 2]) / P[1, 1])[:, None]
    a = dx * dx + dy * dy + 1.0
    disc = centre_distance ** 2 - a * (centre_distance ** 2 - radius ** 2)
    hit = disc >= 0.0
    t = (centre_distance - np.sqrt(np.where(hit, disc, 0.0))) / a          # view z of the surface = -t
    ndc = (P[2, 2] * (-t) + P[2, 3]) / t
    depth = np.where(hit & (t > 0.0), 0.5 * ndc + 0.5, 1.0)
    return np.ascontiguousarray(np.clip(depth, 0.0, 1.0), dtype=np.float32)


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

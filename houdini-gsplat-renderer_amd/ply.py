"""INRIA 3D-Gaussian-Splatting PLY ingest: what the reference's example scene does
between `file1` (PLY import) and `GSplatSource` -- hip/GSplatPlugin_simpleScene_v001.hip, SURVEY App. D:

    Cd      = 0.28209479177387814 * (f_dc_0, f_dc_1, f_dc_2) + 0.5
    opacity = 1 / (1 + exp(-opacity))
    scale   = exp(scale_0, scale_1, scale_2)
    orient  = normalize(rot_1, rot_2, rot_3, rot_0)        (x, y, z, w)
    sh{k}   = (f_rest_{k-1}, f_rest_{k-1+15}, f_rest_{k-1+30}),  k = 1..15
    everything except P cast to fpreal16

The result is a `scenes.Splats` in the registerUpdate() layout, ready for `Engine.upload` or
`GSplatRenderer.registerUpdate`.  Only binary_little_endian / ascii PLY with float properties.
"""
from __future__ import annotations

import numpy as np

from .scenes import Splats, f16bits

SH_C0 = 0.28209479177387814
_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def read_ply_vertices(path: str) -> np.ndarray:
    """structured array of the `vertex` element"""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, count, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header not terminated")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties on vertices are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        dt = np.dtype(props)
        if fmt == "binary_little_endian":
            data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        elif fmt == "ascii":
            flat = np.loadtxt(f, dtype=np.float64, max_rows=count).reshape(count, len(props))
            data = np.zeros(count, dtype=dt)
            for k, (name, _) in enumerate(props):
                data[name] = flat[:, k]
        else:
            raise ValueError(f"unsupported PLY format {fmt}")
    return data


def splats_from_inria(v: np.ndarray, cd_override=None) -> Splats:
    """apply the example scene's activations to raw INRIA attributes"""
    n = v.shape[0]
    names = v.dtype.names
    P = np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float32)
    if cd_override is not None:                       # the example scene overwrites Cd with 0.5 grey (App. D)
        cd = np.broadcast_to(np.asarray(cd_override, np.float32), (n, 3))
    else:
        cd = SH_C0 * np.stack([v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]], axis=1).astype(np.float32) + np.float32(0.5)
    opacity = (1.0 / (1.0 + np.exp(-v["opacity"].astype(np.float32)))).astype(np.float32)
    scale = np.exp(np.stack([v["scale_0"], v["scale_1"], v["scale_2"]], axis=1).astype(np.float32))
    q = np.stack([v["rot_1"], v["rot_2"], v["rot_3"], v["rot_0"]], axis=1).astype(np.float32)
    nrm = np.linalg.norm(q, axis=1, keepdims=True)
    q = np.where(nrm > 0, q / np.where(nrm > 0, nrm, 1), q)
    shx = shy = shz = None
    n_rest = sum(1 for k in range(45) if f"f_rest_{k}" in names)
    if n_rest == 45:                                  # all three degrees present (src/GR_GSplat.C:166-177)
        fr = np.stack([v[f"f_rest_{k}"] for k in range(45)], axis=1).astype(np.float32)
        shx = np.zeros((n, 16), np.uint16)
        shy = np.zeros((n, 16), np.uint16)
        shz = np.zeros((n, 16), np.uint16)
        shx[:, :15] = f16bits(fr[:, 0:15])
        shy[:, :15] = f16bits(fr[:, 15:30])
        shz[:, :15] = f16bits(fr[:, 30:45])
    return Splats(P, f16bits(cd), opacity, f16bits(scale), f16bits(q), shx, shy, shz)


def load_inria_ply(path: str, cd_override=None) -> Splats:
    return splats_from_inria(read_ply_vertices(path), cd_override)

"""Gaussian point-cloud PLY files (second half).

The reference stores / restores a trained model with ``GaussianModel.save_ply`` / ``load_ply``
(gaussian_splatting/scene/gaussian_model.py:178-256) through the third-party ``plyfile`` package: one
``vertex`` element, every property ``float`` (f4), binary little endian, in the order
``x y z nx ny nz f_dc_0..2 f_rest_0..(3(D+1)^2-4) opacity scale_0.. rot_0..`` (:178-190), with the SH features
stored channel-major (``features.transpose(1, 2).flatten(start_dim=1)``, :197-198) and opacity / scale /
rotation as the RAW (pre-activation) parameters.  This module reads and writes that very layout with numpy
only -- host-side I/O, nothing on the GPU -- so that checkpoints of the reference and of this package are
interchangeable.  ``plyfile`` itself is absent here (no network): the format is pinned by the PLY specification
and by the attribute list above, not by a byte comparison with that package (tests/test_ply_cpu.py).
"""
from __future__ import annotations

import os

import numpy as np
import torch

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def attribute_names(n_dc: int, n_rest: int, n_scale: int, n_rot: int):
    """construct_list_of_attributes (gaussian_model.py:178-190)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def save_gaussians_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """save_ply (gaussian_model.py:192-208).  xyz [P,3]; features_dc [P,1,3]; features_rest [P,K-1,3]; opacity [P,1];
    scaling [P,3]; rotation [P,4] -- the model's raw parameters, any device."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    t = lambda x: x.detach().to("cpu", torch.float32)
    xyz = t(xyz).numpy()
    f_dc = t(features_dc).transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    f_rest = t(features_rest).transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    cols = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, t(opacity).reshape(len(xyz), -1).numpy(),
                           t(scaling).numpy(), t(rotation).numpy()], axis=1).astype("<f4")
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], scaling.shape[1], rotation.shape[1])
    assert cols.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % len(xyz)
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(cols).tobytes())


def read_ply_vertices(path):
    """-> dict name -> float64 column of the first element (ascii or binary little / big endian, scalar properties)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first, n_elements = None, 0, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                n_elements += 1
                in_first = n_elements == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=count, ndmin=2)
            return {n: data[:, i].astype(np.float64) for i, (n, _) in enumerate(props)}
        end = "<" if fmt == "binary_little_endian" else ">" if fmt == "binary_big_endian" else None
        if end is None:
            raise ValueError(f"{path}: unknown PLY format {fmt}")
        dt = np.dtype([(n, end + ty) for n, ty in props])
        rec = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        return {n: rec[n].astype(np.float64) for n, _ in props}


def load_gaussians_ply(path, max_sh_degree: int = 3, device="cpu"):
    """load_ply (gaussian_model.py:215-256) -> dict(xyz [P,3], features_dc [P,1,3], features_rest [P,K-1,3],
    opacity [P,1], scaling [P,3], rotation [P,4]) float32 on `device`."""
    v = read_ply_vertices(path)
    by_index = lambda prefix: sorted((k for k in v if k.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))
    xyz = np.stack([v["x"], v["y"], v["z"]], axis=1)
    P = xyz.shape[0]
    f_dc = np.stack([v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]], axis=1).reshape(P, 3, 1)
    rest = by_index("f_rest_")
    if len(rest) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {len(rest)} f_rest_* properties, SH degree {max_sh_degree} needs {3 * (max_sh_degree + 1) ** 2 - 3}")
    f_rest = np.stack([v[k] for k in rest], axis=1).reshape(P, 3, (max_sh_degree + 1) ** 2 - 1) if rest else np.zeros((P, 3, 0))
    to = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
    return dict(xyz=to(xyz), features_dc=to(f_dc).transpose(1, 2).contiguous(), features_rest=to(f_rest).transpose(1, 2).contiguous(),
                opacity=to(v["opacity"][:, None]), scaling=to(np.stack([v[k] for k in by_index("scale_")], axis=1)),
                rotation=to(np.stack([v[k] for k in by_index("rot")], axis=1)))

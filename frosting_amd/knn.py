"""``distCUDA2`` -- mean squared distance to the three nearest neighbours.

Drop-in for ``from simple_knn._C import distCUDA2`` (gaussian_splatting/scene/gaussian_model.py:20,134;
frosting_scene/frosting_model.py:9,530; frosting_scene/sugar_model.py:9): ``distCUDA2(points [P,3] float32
cuda) -> [P] float32``.  ``install_as_simple_knn()`` registers a module of that name so the reference's
import line works unchanged.  GPU only.
"""
from __future__ import annotations

import ctypes as C
import sys
import types

import torch

from . import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if points.device.type != "cuda":
        raise RuntimeError("frosting_amd distCUDA2 runs on the GPU only (no CPU path)")
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    pts = points.contiguous().float()
    P = int(pts.shape[0])
    dev = pts.device
    out = torch.full((P,), 0.0, dtype=torch.float32, device=dev)       # spatial.cu:22
    if P == 0:
        return out
    L = _lib.lib()
    with torch.cuda.device(dev):
        ws = torch.empty(int(L.frg_knn_workspace_bytes(P)) + 256, dtype=torch.uint8, device=dev)
        base = (ws.data_ptr() + 255) // 256 * 256
        stream = torch.cuda.current_stream(dev)
        rc = L.frg_knn_mean_dist2(P, C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(base),
                                  ws.numel() - (base - ws.data_ptr()), C.c_void_p(stream.cuda_stream))
        ws.record_stream(stream)
    if rc < 0:
        raise RuntimeError(f"frg_knn_mean_dist2 failed ({rc}): {_lib.last_error()}")
    return out


def install_as_simple_knn():
    """Make ``from simple_knn._C import distCUDA2`` resolve to this implementation."""
    pkg, sub = types.ModuleType("simple_knn"), types.ModuleType("simple_knn._C")
    sub.distCUDA2 = distCUDA2
    pkg._C = sub
    sys.modules["simple_knn"], sys.modules["simple_knn._C"] = pkg, sub
    return sub

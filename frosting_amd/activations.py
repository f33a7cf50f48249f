"""Gaussian parameter activations (part of SURVEY.md 8(f) rank 3).

``opacity = sigmoid(raw)``, ``scale = exp(raw)``, ``rotation = normalize(raw)`` -- the reference's
``get_opacity`` / ``get_scaling`` / ``get_rotation`` (gaussian_splatting/scene/gaussian_model.py:32-40,96-115)
and ``strengths`` / ``scaling`` / ``quaternions`` (frosting_scene/frosting_model.py:726-798, non-editable
case) -- as one launch forward and one launch backward.  GPU only.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _check(P, *tensors):
    dev = tensors[0].device
    if dev.type != "cuda":
        raise RuntimeError("frosting_amd activations run on the GPU only (no CPU path)")
    for t in tensors:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev or t.shape[0] != P:
            raise RuntimeError("expected contiguous float32 tensors with one row per Gaussian on one device")
    return dev


def activate(raw_opacity, raw_scale, raw_rot, out=None):
    """-> (opacity [P,1], scale [P,3], rotation [P,4]); `out` = optional preallocated triple."""
    P = raw_opacity.shape[0]
    dev = _check(P, raw_opacity, raw_scale, raw_rot)
    o, s, r = out if out is not None else (torch.empty_like(raw_opacity), torch.empty_like(raw_scale), torch.empty_like(raw_rot))
    rc = _lib.lib().frg_activate(P, _ptr(raw_opacity), _ptr(raw_scale), _ptr(raw_rot), _ptr(o), _ptr(s), _ptr(r),
                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc < 0:
        raise RuntimeError(f"frg_activate failed ({rc}): {_lib.last_error()}")
    return o, s, r


def activate_backward_(opacity, scale, raw_rot, g_opacity, g_scale, g_rot):
    """In place: gradients w.r.t. the activated values -> gradients w.r.t. the raw parameters."""
    P = opacity.shape[0]
    dev = _check(P, opacity, scale, raw_rot, g_opacity, g_scale, g_rot)
    rc = _lib.lib().frg_activate_backward(P, _ptr(opacity), _ptr(scale), _ptr(raw_rot), _ptr(g_opacity), _ptr(g_scale),
                                          _ptr(g_rot), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc < 0:
        raise RuntimeError(f"frg_activate_backward failed ({rc}): {_lib.last_error()}")
    return g_opacity, g_scale, g_rot


class _Activations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_opacity, raw_scale, raw_rot):
        ro, rs, rr = raw_opacity.contiguous(), raw_scale.contiguous(), raw_rot.contiguous()
        o, s, r = activate(ro, rs, rr)
        ctx.save_for_backward(o, s, rr)
        return o, s, r

    @staticmethod
    def backward(ctx, go, gs, gr):
        o, s, rr = ctx.saved_tensors
        return activate_backward_(o, s, rr, go.contiguous().clone(), gs.contiguous().clone(), gr.contiguous().clone())


def gaussian_activations(raw_opacity, raw_scale, raw_rot):
    """Differentiable (opacity, scale, rotation) = (sigmoid, exp, normalize)(raw)."""
    return _Activations.apply(raw_opacity, raw_scale, raw_rot)


# ---- Frosting's shell parameterisation of the centres (frosting_model.py:713-724) -------------------
class _ShellPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bary_logits, cell_verts, point_cell_indices):
        lg, cv, ci = bary_logits.contiguous(), cell_verts.contiguous(), point_cell_indices.contiguous()
        P = lg.shape[0]
        dev = _check(P, lg)
        if lg.shape[1:] != (6,) or cv.dtype != torch.float32 or cv.device != dev or cv.reshape(-1).numel() % 18 or \
                ci.dtype != torch.int64 or ci.device != dev or ci.shape != (P,):
            raise RuntimeError("expected bary_logits [P,6] float32, cell_verts [F,6,3] (or [F,2,3,3]) float32, "
                               "point_cell_indices [P] int64 on one GPU")
        pts = torch.empty((P, 3), dtype=torch.float32, device=dev)
        rc = _lib.lib().frg_shell_points(P, _ptr(lg), _ptr(cv), _ptr(ci), _ptr(pts),
                                         C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc < 0:
            raise RuntimeError(f"frg_shell_points failed ({rc}): {_lib.last_error()}")
        ctx.save_for_backward(lg, cv, ci)
        return pts

    @staticmethod
    def backward(ctx, g):
        lg, cv, ci = ctx.saved_tensors
        out = torch.empty_like(lg)
        rc = _lib.lib().frg_shell_points_backward(lg.shape[0], _ptr(lg), _ptr(cv), _ptr(ci), _ptr(g.contiguous()), _ptr(out),
                                                  C.c_void_p(torch.cuda.current_stream(lg.device).cuda_stream))
        if rc < 0:
            raise RuntimeError(f"frg_shell_points_backward failed ({rc}): {_lib.last_error()}")
        return out, None, None


def shell_points(bary_logits, cell_verts, point_cell_indices):
    """``(softmax(_bary_coords)[..., None] * shell_cells_verts[_point_cell_indices].reshape(-1, 6, 3)).sum(-2)``,
    differentiable w.r.t. the logits (cell vertices are constants: the reference's learn_shell = False)."""
    return _ShellPoints.apply(bary_logits, cell_verts, point_cell_indices)

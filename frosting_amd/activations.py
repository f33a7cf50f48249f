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

"""Fused photometric loss: ``(1 - lambda) * L1 + lambda * (1 - SSIM)``.

Host mirror of ``frosting_utils/loss_utils.py:17-62`` as ``frosting_trainers/refine.py:407-409``
combines it (``dssim_factor = 0.2``): same window (11 taps, sigma 1.5, built in float32 the way
``gaussian()`` / ``create_window()`` build it), same constants, same zero padding.  Forward and
backward are two tiled HIP kernels behind ``frg_photometric_loss``; the gradient with respect to the
rendered image is what the rasterizer's backward takes as ``dL_dout_color``.  GPU only.
"""
from __future__ import annotations

import ctypes as C
from math import exp

import torch

from . import _lib

WINDOW_SIZE = 11
SIGMA = 1.5


def gaussian_window(window_size: int = WINDOW_SIZE, sigma: float = SIGMA) -> torch.Tensor:
    """loss_utils.py:23-25, float32 like the reference's torch.Tensor([...]) / sum()."""
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


_WINDOW = None


def _window_c():
    global _WINDOW
    if _WINDOW is None:
        w = gaussian_window()
        _WINDOW = (C.c_float * WINDOW_SIZE)(*[float(v) for v in w])
    return _WINDOW


def photometric_loss_and_grad(image: torch.Tensor, target: torch.Tensor, lambda_dssim: float = 0.2, need_grad: bool = True):
    """image, target: [C,H,W] (or [1,C,H,W]) float32 on the GPU.  Returns (loss [1] device tensor,
    d loss / d image or None)."""
    if image.device.type != "cuda":
        raise RuntimeError("frosting_amd photometric loss runs on the GPU only (no CPU path)")
    if image.shape != target.shape or image.dtype != torch.float32 or target.dtype != torch.float32 or target.device != image.device:
        raise RuntimeError("image and target must be float32 tensors of the same shape on the same device")
    shape = image.shape
    if image.dim() == 4 and shape[0] == 1:
        image, target = image[0], target[0]
    if image.dim() != 3:
        raise RuntimeError("expected [C,H,W] or [1,C,H,W]")
    img, tgt = image.contiguous(), target.contiguous()
    Cn, H, W = (int(v) for v in img.shape)
    L = _lib.lib()
    dev = img.device
    with torch.cuda.device(dev):
        ws = torch.empty(int(L.frg_photometric_workspace_bytes(Cn, W, H)), dtype=torch.uint8, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        grad = torch.empty_like(img) if need_grad else None
        rc = L.frg_photometric_loss(Cn, W, H, C.c_void_p(img.data_ptr()), C.c_void_p(tgt.data_ptr()), _window_c(),
                                    float(lambda_dssim), C.c_void_p(loss.data_ptr()),
                                    C.c_void_p(grad.data_ptr()) if need_grad else None,
                                    C.c_void_p(ws.data_ptr()), ws.numel(),
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        ws.record_stream(torch.cuda.current_stream(dev))
    if rc < 0:
        raise RuntimeError(f"frg_photometric_loss failed ({rc}): {_lib.last_error()}")
    return loss, (grad.view(shape) if need_grad else None)


class _PhotometricLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, target, lambda_dssim):
        loss, grad = photometric_loss_and_grad(image, target, lambda_dssim, need_grad=image.requires_grad)
        ctx.save_for_backward(grad if grad is not None else torch.empty(0))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g if grad.numel() else None), None, None


def photometric_loss(image: torch.Tensor, target: torch.Tensor, lambda_dssim: float = 0.2) -> torch.Tensor:
    """Differentiable w.r.t. image: drop-in for ``(1 - f) * l1_loss(pred, gt) + f * (1.0 - ssim(pred, gt))``."""
    return _PhotometricLoss.apply(image, target, lambda_dssim)

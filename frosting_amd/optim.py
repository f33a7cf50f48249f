"""Fused Adam over the flat per-Gaussian parameter layout.

The reference steps ``torch.optim.Adam(groups, lr=0.0, eps=1e-15)`` with one parameter group per
tensor and per-group learning rates (frosting_scene/frosting_optimizer.py:74-121,
gaussian_splatting/scene/gaussian_model.py:149-167) -- half a dozen eager elementwise kernels per
group.  Here parameters, both moments and the gradients share one flat fp32 layout (the gradient
side IS the exchange buffer of frosting_amd.parallel.GradientExchange, so the summed gradients are
consumed where the all-reduce left them) and ``frg_adam_step`` updates every group in one launch,
28 bytes of HBM traffic per element.  GPU only: there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .parallel import PARAM_ORDER


class FlatAdam:
    """``FlatAdam(shapes, lrs, device)``: ``params[name]`` are views into one flat buffer (write the
    initial values into them); ``step(flat_grads)`` applies one Adam update to all of them.  Same
    update rule, defaults (betas 0.9/0.999) and eps handling as ``torch.optim.Adam`` without weight
    decay / amsgrad."""

    def __init__(self, shapes: dict, lrs: dict, device, betas=(0.9, 0.999), eps: float = 1e-15, sh_dc_lr=None):
        """sh_dc_lr: learning rate of the DC coefficient of "shs" ([P,K,3]); lrs["shs"] then applies to the
        other K-1 coefficients -- the reference's features_dc / features_rest groups on one tensor."""
        self.device = torch.device(device)
        names = [k for k in PARAM_ORDER if k in shapes] + [k for k in shapes if k not in PARAM_ORDER]
        if not 1 <= len(names) <= 8:
            raise ValueError("1..8 parameter groups expected")
        self.names = names
        self.shapes = {k: tuple(shapes[k]) for k in names}
        sizes = [int(torch.Size(self.shapes[k]).numel()) for k in names]
        self.numel = sum(sizes)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.params, ends, o = {}, [], 0
        for k, n in zip(names, sizes):
            self.params[k] = self.flat[o:o + n].view(self.shapes[k])
            o += n
            ends.append(o)
        self._ends = (C.c_longlong * len(names))(*ends)
        self.lrs = {k: float(lrs[k]) for k in names}
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.steps = 0
        self.sh_dc_lr = None if sh_dc_lr is None else float(sh_dc_lr)
        n = len(names)
        self._period, self._head = (C.c_int * n)(*([0] * n)), (C.c_int * n)(*([0] * n))
        if self.sh_dc_lr is not None:
            if "shs" not in self.shapes or len(self.shapes["shs"]) != 3:
                raise ValueError('sh_dc_lr needs a "shs" group of shape [P,K,3]')
            k = names.index("shs")
            self._period[k], self._head[k] = int(self.shapes["shs"][1] * self.shapes["shs"][2]), int(self.shapes["shs"][2])

    def set_lr(self, name: str, lr: float):
        """Per-group learning-rate schedule hook (reference: update_learning_rate)."""
        if name not in self.lrs:
            raise KeyError(name)
        self.lrs[name] = float(lr)

    def step(self, flat_grads: torch.Tensor, grad_scale: float = 1.0):
        g = flat_grads
        if g.device.type != "cuda" or self.flat.device.type != "cuda":
            raise RuntimeError("FlatAdam runs on the GPU only (no CPU path)")
        if g.dtype != torch.float32 or g.numel() != self.numel or not g.is_contiguous() or g.device != self.flat.device:
            raise RuntimeError(f"expected a contiguous float32 gradient buffer of {self.numel} elements on {self.flat.device}")
        self.steps += 1
        lrs = (C.c_float * len(self.names))(*[self.lrs[k] for k in self.names])
        head_lrs = (C.c_float * len(self.names))(*[(self.sh_dc_lr if (k == "shs" and self.sh_dc_lr is not None) else 0.0)
                                                    for k in self.names])
        stream = C.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)
        rc = _lib.lib().frg_adam_step(self.numel, C.c_void_p(self.flat.data_ptr()), C.c_void_p(g.data_ptr()),
                                      C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()),
                                      self._ends, lrs, self._period, self._head, head_lrs, len(self.names),
                                      self.betas[0], self.betas[1], self.eps,
                                      self.steps, float(grad_scale), stream)
        if rc < 0:
            raise RuntimeError(f"frg_adam_step failed ({rc}): {_lib.last_error()}")
        return self.params

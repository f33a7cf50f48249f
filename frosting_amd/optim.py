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
from .parallel import PARAM_ORDER, flat_layout


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
        self.lrs = {k: float(lrs[k]) for k in names}
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.steps = 0
        self.sh_dc_lr = None if sh_dc_lr is None else float(sh_dc_lr)
        self._build({k: tuple(shapes[k]) for k in names})

    def _build(self, shapes: dict, old=None):
        """(Re)allocate the flat buffers for `shapes`; `old` = (params, exp_avg views, exp_avg_sq views,
        row selector per name) carries state over (densification / pruning)."""
        names = self.names
        self.shapes = shapes
        _, self.layout, self.numel = flat_layout(shapes, names)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        view = lambda buf: {k: buf[o:o + n].view(shapes[k]) for k, (o, n) in self.layout.items()}
        self.params, self.m, self.v = view(self.flat), view(self.exp_avg), view(self.exp_avg_sq)
        # segment k ends where segment k+1 starts (its alignment pad rides along with it, all zeros)
        ends = [self.layout[names[i + 1]][0] if i + 1 < len(names) else self.numel for i in range(len(names))]
        self._ends = (C.c_longlong * len(names))(*ends)
        n = len(names)
        self._period, self._head = (C.c_int * n)(*([0] * n)), (C.c_int * n)(*([0] * n))
        if self.sh_dc_lr is not None:
            if "shs" not in shapes or len(shapes["shs"]) != 3:
                raise ValueError('sh_dc_lr needs a "shs" group of shape [P,K,3]')
            k = names.index("shs")
            self._period[k], self._head[k] = int(shapes["shs"][1] * shapes["shs"][2]), int(shapes["shs"][2])
        if old is not None:
            for k in names:
                for dst, src in ((self.params, old[0]), (self.m, old[1]), (self.v, old[2])):
                    rows = src[k] if old[3] is None else src[k][old[3]]
                    dst[k][: rows.shape[0]].copy_(rows)

    # ---- densification / pruning (reference: gaussian_model.py _prune_optimizer, cat_tensors_to_optimizer,
    # replace_tensor_to_optimizer; frosting_optimizer.py keeps the same per-group state) -------------------
    def _per_gaussian(self):
        P = {self.shapes[k][0] for k in self.names}
        if len(P) != 1:
            raise RuntimeError("prune / append need every group to be per-Gaussian (same leading dimension)")
        return P.pop()

    def prune(self, keep_mask: torch.Tensor):
        """Keep the Gaussians with a true entry: parameters and both moments of every group are
        compacted consistently (gaussian_model.py:_prune_optimizer).  Returns the new params dict."""
        P = self._per_gaussian()
        keep = keep_mask.to(self.device).reshape(-1).bool()
        if keep.numel() != P:
            raise RuntimeError(f"keep_mask has {keep.numel()} entries, the model has {P} Gaussians")
        n = int(keep.sum())
        old = (self.params, self.m, self.v, keep)
        self._build({k: (n,) + tuple(self.shapes[k][1:]) for k in self.names}, old)
        return self.params

    def append(self, new_params: dict):
        """Add Gaussians: new_params[name] is [n, ...]; their moments start at zero
        (gaussian_model.py:cat_tensors_to_optimizer).  Returns the new params dict."""
        P = self._per_gaussian()
        n = {int(new_params[k].shape[0]) for k in self.names}
        if len(n) != 1:
            raise RuntimeError("every group needs the same number of new rows")
        n = n.pop()
        old = (self.params, self.m, self.v, None)
        self._build({k: (P + n,) + tuple(self.shapes[k][1:]) for k in self.names}, old)
        for k in self.names:
            self.params[k][P:].copy_(new_params[k].to(self.device))
        return self.params

    def reset(self, name: str, values: torch.Tensor = None, rows=None):
        """Overwrite a group (or some of its rows) and zero the matching moments
        (gaussian_model.py:replace_tensor_to_optimizer, used by the opacity reset)."""
        if name not in self.params:
            raise KeyError(name)
        sel = slice(None) if rows is None else rows
        if values is not None:
            self.params[name][sel] = values.to(self.device)
        self.m[name][sel] = 0.0
        self.v[name][sel] = 0.0

    def set_lr(self, name: str, lr: float):
        """Per-group learning-rate schedule hook (reference: update_learning_rate)."""
        if name not in self.lrs:
            raise KeyError(name)
        self.lrs[name] = float(lr)

    def step(self, flat_grads: torch.Tensor, grad_scale: float = 1.0, row_live: torch.Tensor = None):
        """row_live (optional, uint8 [P], as frg_backward_args::row_live leaves it): the gradient rows of unmarked Gaussians
        were never written -- they count as zero and are not read (the moments decay, the parameter moves by its momentum:
        what a stored zero gives, bit for bit).  Every group must then be per-Gaussian."""
        g = flat_grads
        if g.device.type != "cuda" or self.flat.device.type != "cuda":
            raise RuntimeError("FlatAdam runs on the GPU only (no CPU path)")
        if g.dtype != torch.float32 or g.numel() != self.numel or not g.is_contiguous() or g.device != self.flat.device:
            raise RuntimeError(f"expected a contiguous float32 gradient buffer of {self.numel} elements on {self.flat.device}")
        if row_live is None and getattr(g, "_frg_rows_partial", False):
            raise RuntimeError("this gradient buffer was written in live_rows mode (only the rows of Gaussians with a gradient): "
                               "pass row_live=, or ViewParallelRasterizer.zero_dead_rows() first")
        lrs = (C.c_float * len(self.names))(*[self.lrs[k] for k in self.names])
        head_lrs = (C.c_float * len(self.names))(*[(self.sh_dc_lr if (k == "shs" and self.sh_dc_lr is not None) else 0.0)
                                                    for k in self.names])
        stream = C.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)
        if row_live is not None:
            P = self._per_gaussian()
            if row_live.dtype != torch.uint8 or row_live.numel() != P or row_live.device != self.flat.device or not row_live.is_contiguous():
                raise RuntimeError(f"row_live: expected a contiguous uint8 tensor of {P} entries on {self.flat.device}")
            widths = (C.c_int * len(self.names))(*[int(torch.Size(self.shapes[k][1:]).numel()) for k in self.names])
            rc = _lib.lib().frg_adam_step_rows(self.numel, C.c_void_p(self.flat.data_ptr()), C.c_void_p(g.data_ptr()),
                                               C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()),
                                               self._ends, lrs, self._period, self._head, head_lrs, len(self.names),
                                               self.betas[0], self.betas[1], self.eps, self.steps + 1, float(grad_scale),
                                               C.c_void_p(row_live.data_ptr()), P, widths, stream)
        else:
            rc = _lib.lib().frg_adam_step(self.numel, C.c_void_p(self.flat.data_ptr()), C.c_void_p(g.data_ptr()),
                                          C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()),
                                          self._ends, lrs, self._period, self._head, head_lrs, len(self.names),
                                          self.betas[0], self.betas[1], self.eps,
                                          self.steps + 1, float(grad_scale), stream)
        if rc < 0:
            raise RuntimeError(f"frg_adam_step failed ({rc}): {_lib.last_error()}")
        self.steps += 1                       # only a step that ran advances the bias correction
        return self.params


class ShardedFlatAdam(FlatAdam):
    """SURVEY.md 8(e), second option: instead of an all-reduce of the gradients and a replicated update, the gradients are
    REDUCE-SCATTERED, every rank updates its 1/N shard of the flat parameter buffer (both moments exist only for the
    shard), and the updated shards are ALL-GATHERED: the same wire bytes as the all-reduce's two halves, a 1/N Adam
    (0.8 ms of a 2.4 ms training step at C3).  ``step(flat_grads)`` takes THIS rank's (per-view) gradient buffer; the
    parameters it leaves in ``self.params`` are those of FlatAdam.step on the sum over ranks, bit for bit (the sum of a
    shard is taken in rank order on backends without a reduce-scatter).

    Whether it pays depends on the node: the parameter all-gather moves (N-1)/N x 708 MB at 3 M Gaussians -- 0.58 ms at the
    nominal 7 x 153 GB/s, 1.4 ms at 450 GB/s -- against the 0.7 ms it takes off the replicated update
    (frosting_amd.parallel.predict_exchange reports both)."""

    def __init__(self, shapes: dict, lrs: dict, device, process_group, betas=(0.9, 0.999), eps: float = 1e-15, sh_dc_lr=None,
                 shard_step=None):
        import torch.distributed as dist
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self._shard_step = shard_step           # tests on CPU tensors inject the update of one shard (the product one is HIP-only)
        super().__init__(shapes, lrs, device, betas=betas, eps=eps, sh_dc_lr=sh_dc_lr)

    def _build(self, shapes: dict, old=None):
        if old is not None:
            raise RuntimeError("ShardedFlatAdam: prune / append re-shard the moments -- not implemented; rebuild the optimizer")
        names = self.names
        self.shapes = shapes
        _, self.layout, self.numel = flat_layout(shapes, names)
        # shards of equal length, a multiple of four elements (16-byte accesses); the flat buffers are padded to N shards
        self.shard = ((self.numel + self.world - 1) // self.world + 3) // 4 * 4
        padded = self.shard * self.world
        self.flat_padded = torch.zeros(padded, dtype=torch.float32, device=self.device)
        self.flat = self.flat_padded[: self.numel]
        self.exp_avg = torch.zeros(self.shard, dtype=torch.float32, device=self.device)        # this rank's shard only
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.params = {k: self.flat[o:o + n].view(shapes[k]) for k, (o, n) in self.layout.items()}
        ends = [self.layout[names[i + 1]][0] if i + 1 < len(names) else self.numel for i in range(len(names))]
        ends[-1] = padded                                  # the pad behind the last segment rides with it (zeros: never moves)
        self._ends = (C.c_longlong * len(names))(*ends)
        n = len(names)
        self._period, self._head = (C.c_int * n)(*([0] * n)), (C.c_int * n)(*([0] * n))
        if self.sh_dc_lr is not None:
            k = names.index("shs")
            self._period[k], self._head[k] = int(shapes["shs"][1] * shapes["shs"][2]), int(shapes["shs"][2])
        self._grad_padded = torch.zeros(padded, dtype=torch.float32, device=self.device)
        self._grad_shard = torch.zeros(self.shard, dtype=torch.float32, device=self.device)
        self._recv = None

    def step(self, flat_grads: torch.Tensor, grad_scale: float = 1.0):
        import torch.distributed as dist
        g = flat_grads
        if g.dtype != torch.float32 or g.numel() != self.numel or g.device != self.flat.device:
            raise RuntimeError(f"expected a float32 gradient buffer of {self.numel} elements on {self.flat.device}")
        self._grad_padded[: self.numel].copy_(g.reshape(-1))
        lo = self.rank * self.shard
        # 1. this rank's shard of the SUM of the gradients
        if dist.get_backend(self.group) == "nccl":
            dist.reduce_scatter_tensor(self._grad_shard, self._grad_padded, op=dist.ReduceOp.SUM, group=self.group)
        else:       # gloo: all-to-all of the shards + a sum in rank order
            if self._recv is None:
                self._recv = torch.empty((self.world, self.shard), dtype=torch.float32, device=self.device)
            dist.all_to_all_single(self._recv.view(-1), self._grad_padded, group=self.group)
            torch.sum(self._recv, dim=0, out=self._grad_shard)
        # 2. the update of the shard
        mine = self.flat_padded[lo: lo + self.shard]
        lrs = [self.lrs[k] for k in self.names]
        head_lrs = [(self.sh_dc_lr if (k == "shs" and self.sh_dc_lr is not None) else 0.0) for k in self.names]
        if self._shard_step is not None:
            self._shard_step(self, lo, mine, self._grad_shard, lrs, head_lrs, float(grad_scale))
        else:
            if self.flat.device.type != "cuda":
                raise RuntimeError("ShardedFlatAdam runs on the GPU only (no CPU path; tests inject their own shard update)")
            n = len(self.names)
            stream = C.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)
            rc = _lib.lib().frg_adam_step_shard(self.shard, lo, C.c_void_p(mine.data_ptr()), C.c_void_p(self._grad_shard.data_ptr()),
                                                C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()),
                                                self._ends, (C.c_float * n)(*lrs), self._period, self._head, (C.c_float * n)(*head_lrs), n,
                                                self.betas[0], self.betas[1], self.eps, self.steps + 1, float(grad_scale), stream)
            if rc < 0:
                raise RuntimeError(f"frg_adam_step_shard failed ({rc}): {_lib.last_error()}")
        # 3. every rank's updated shard to every rank
        dist.all_gather_into_tensor(self.flat_padded, mine.clone(), group=self.group)
        self.steps += 1
        return self.params

    def reset(self, name: str, values: torch.Tensor = None, rows=None):
        """FlatAdam.reset on the sharded state (gaussian_model.py:replace_tensor_to_optimizer, the opacity reset): the
        parameter rows are overwritten on EVERY rank (they are replicated: every rank calls this with the same arguments),
        the moments only where the group's elements meet this rank's shard [rank * shard, (rank + 1) * shard)."""
        if name not in self.params:
            raise KeyError(name)
        sel = slice(None) if rows is None else rows
        if values is not None:
            self.params[name][sel] = values.to(self.device)
        off, n = self.layout[name]
        lo = self.rank * self.shard
        if rows is None:
            a, b = max(off, lo), min(off + n, lo + self.shard)
            if a < b:
                self.exp_avg[a - lo: b - lo] = 0.0
                self.exp_avg_sq[a - lo: b - lo] = 0.0
            return
        # flat positions of the selected rows' elements, those inside the shard shifted to its origin
        idx = torch.arange(n, device=self.device).view(self.shapes[name])[sel].reshape(-1) + off
        idx = idx[(idx >= lo) & (idx < lo + self.shard)] - lo
        if idx.numel():
            self.exp_avg.index_fill_(0, idx, 0.0)
            self.exp_avg_sq.index_fill_(0, idx, 0.0)

    def prune(self, keep_mask):
        raise RuntimeError("ShardedFlatAdam: prune / append re-shard the moments -- not implemented; rebuild the optimizer")

    append = prune

"""Rasterization straight from the model's RAW parameters.

Every iteration the reference turns its optimizer's parameters into rasterizer inputs with eager torch ops --
``sigmoid(_opacities)``, ``exp(_scales)``, ``F.normalize(_quaternions)`` and, in Frosting,
``(softmax(_bary_coords)[..., None] * shell_cells_verts[_point_cell_indices].reshape(-1, 6, 3)).sum(-2)`` for the
shell-bound centres (frosting_scene/frosting_model.py:707-798, 1498-1520; gaussian_splatting/scene/gaussian_model.py:96-115)
-- and autograd replays the chain backwards.  ``rasterize_raw`` hands the raw tensors to the rasterizer instead:
the activations run inside the per-Gaussian kernels (csrc/raw_params.h), no activated tensor is ever written,
and the backward returns gradients with respect to the raw parameters, the barycentric logits and -- when the
cell vertices require grad (``learn_shell = True``) -- the shell itself.  GPU only.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .parallel import _Arena


def _p(t):
    return None if t is None else t.data_ptr()


def _f32(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


class _RasterizeRaw(torch.autograd.Function):
    # tensor inputs, in the order of forward()'s arguments after `settings` and `opts`
    NAMES = ("shs", "raw_opacity", "raw_scale", "raw_rot", "means3D", "shell_logits", "shell_cell_verts", "shell_cells",
             "keep_mask", "means2D")

    @staticmethod
    def _views(tensors):
        """contiguous float32 / int64 views of the saved inputs, as the C ABI wants them"""
        d = dict(zip(_RasterizeRaw.NAMES, tensors))
        return dict(shs=_f32(d["shs"]), ro=_f32(d["raw_opacity"]).reshape(-1), rs=_f32(d["raw_scale"]), rr=_f32(d["raw_rot"]),
                    means=_f32(d["means3D"]), lg=_f32(d["shell_logits"]),
                    cv=None if d["shell_cell_verts"] is None else _f32(d["shell_cell_verts"]).reshape(-1, 6, 3),
                    ci=None if d["shell_cells"] is None else d["shell_cells"].to(torch.int64).contiguous(),
                    mask=None if d["keep_mask"] is None else d["keep_mask"].contiguous())

    @staticmethod
    def forward(ctx, settings, opts, shs, raw_opacity, raw_scale, raw_rot, means3D, shell_logits, shell_cell_verts, shell_cells,
                keep_mask, means2D):
        L = _lib.lib()
        s = settings
        shelled = shell_logits is not None
        ref = shell_logits if shelled else means3D
        dev = ref.device
        if dev.type != "cuda":
            raise RuntimeError("frosting_amd.fused: parameters must live on a ROCm device (no CPU path)")
        P, H, W = int(ref.shape[0]), int(s.image_height), int(s.image_width)
        inputs = (shs, raw_opacity, raw_scale, raw_rot, means3D, shell_logits, shell_cell_verts, shell_cells, keep_mask, means2D)
        t = _RasterizeRaw._views(inputs)
        cam = dict(bg=_f32(s.bg), view=_f32(s.viewmatrix), proj=_f32(s.projmatrix), campos=_f32(s.campos))
        modes = _lib.mode_fields(opts.get("modes"))
        if not any(ctx.needs_input_grad):      # no parameter needs a gradient (or torch.no_grad()): nothing is kept for a backward
            modes["forward_only"] = 1
        bary_mode = 0 if opts.get("use_softmax_for_bary_coords", True) else 1
        with torch.cuda.device(dev):
            color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            geom, binning, img = _Arena(dev, 1.0), _Arena(dev, 1.0), _Arena(dev, 1.0)
            a = _lib.ForwardArgs(
                struct_size=C.sizeof(_lib.ForwardArgs), geometry_alloc=geom.cb, binning_alloc=binning.cb, image_alloc=img.cb,
                user=None, P=P, D=int(s.sh_degree), M=int(t["shs"].shape[1]), background=_p(cam["bg"]), width=W, height=H,
                means3D=_p(t["means"]), shs=_p(t["shs"]), colors_precomp=None, opacities=None, scales=None,
                scale_modifier=float(s.scale_modifier), rotations=None, cov3D_precomp=None, viewmatrix=_p(cam["view"]),
                projmatrix=_p(cam["proj"]), cam_pos=_p(cam["campos"]), tan_fovx=float(s.tanfovx), tan_fovy=float(s.tanfovy),
                prefiltered=int(bool(s.prefiltered)), out_color=color.data_ptr(), radii=radii.data_ptr(), debug=int(bool(s.debug)),
                hip_stream=torch.cuda.current_stream(dev).cuda_stream, instance_capacity=0, keep_mask=_p(t["mask"]),
                raw_opacities=_p(t["ro"]), raw_scales=_p(t["rs"]), raw_rotations=_p(t["rr"]), shell_logits=_p(t["lg"]),
                shell_cell_verts=_p(t["cv"]), shell_cells=_p(t["ci"]), shell_bary_mode=bary_mode, **modes)
            R = L.frg_forward_ex(C.byref(a))
        if R < 0:
            raise RuntimeError(f"frg_forward_ex failed ({R}): {_lib.last_error()}")
        # the INPUT tensors are saved (autograd's version counters then catch an in-place optimizer step between this
        # forward and its backward); their float32 views are re-derived in backward -- free for contiguous float32
        ctx.save_for_backward(*[x for x in inputs if x is not None])
        ctx.present = [x is not None for x in inputs]
        ctx.cam, ctx.settings, ctx.R, ctx.bufs, ctx.radii = cam, s, R, (geom, binning, img), radii
        ctx.bary_mode, ctx.exact = bary_mode, modes["exact_blend"]
        ctx.cv_shape = None if shell_cell_verts is None else tuple(shell_cell_verts.shape)
        ctx.ro_shape = tuple(raw_opacity.shape)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, g_color, _g_radii):
        L = _lib.lib()
        s, R, cam = ctx.settings, ctx.R, ctx.cam
        saved = iter(ctx.saved_tensors)
        t = _RasterizeRaw._views([next(saved) if here else None for here in ctx.present])
        geom, binning, img = ctx.bufs
        dev = g_color.device
        shelled = t["lg"] is not None
        # needs_input_grad: (settings, opts, shs, raw_opacity, raw_scale, raw_rot, means3D, shell_logits, shell_cell_verts, ...)
        learn_shell = shelled and t["cv"] is not None and ctx.needs_input_grad[8]
        P = int((t["lg"] if shelled else t["means"]).shape[0])
        H, W = int(s.image_height), int(s.image_width)
        M = int(t["shs"].shape[1])
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            g = dict(m2=e(P, 3), op=e(P), col=e(P, 3), m3=e(P, 3), cov=e(P, 6), sh=e(P, M, 3), sc=e(P, 3), rot=e(P, 4),
                     lg=e(P, 6) if shelled else None,
                     cv=torch.zeros_like(t["cv"]) if learn_shell else None)
            out = g
            if P:
                ws = int(L.frg_backward_workspace_bytes(P, R))
                work = torch.empty(ws, dtype=torch.uint8, device=dev)
                gp = g_color.detach().to(torch.float32).contiguous()
                a = _lib.BackwardArgs(
                    struct_size=C.sizeof(_lib.BackwardArgs), P=P, D=int(s.sh_degree), M=M, R=R, background=_p(cam["bg"]), width=W,
                    height=H, means3D=_p(t["means"]), shs=_p(t["shs"]), colors_precomp=None, scales=None,
                    scale_modifier=float(s.scale_modifier), rotations=None, cov3D_precomp=None, viewmatrix=_p(cam["view"]),
                    projmatrix=_p(cam["proj"]), campos=_p(cam["campos"]), tan_fovx=float(s.tanfovx), tan_fovy=float(s.tanfovy),
                    radii=ctx.radii.data_ptr(), geom_buffer=geom.buf.data_ptr(), binning_buffer=binning.buf.data_ptr(),
                    image_buffer=img.buf.data_ptr(), dL_dpix=gp.data_ptr(), dL_dmean2D=_p(g["m2"]), dL_dconic=None,
                    dL_dopacity=_p(g["op"]), dL_dcolor=_p(g["col"]), dL_dmean3D=_p(g["m3"]), dL_dcov3D=_p(g["cov"]),
                    dL_dsh=_p(g["sh"]), dL_dscale=_p(g["sc"]), dL_drot=_p(g["rot"]), workspace=work.data_ptr(), workspace_bytes=ws,
                    debug=int(bool(s.debug)), hip_stream=torch.cuda.current_stream(dev).cuda_stream,
                    raw_opacities=_p(t["ro"]), raw_scales=_p(t["rs"]), raw_rotations=_p(t["rr"]), shell_logits=_p(t["lg"]),
                    shell_cell_verts=_p(t["cv"]), shell_cells=_p(t["ci"]), dL_dshell_logits=_p(g["lg"]),
                    dL_dshell_cell_verts=_p(g["cv"]), exact_blend=ctx.exact, shell_bary_mode=ctx.bary_mode)
                rc = L.frg_backward_ex(C.byref(a))
                if rc < 0:
                    raise RuntimeError(f"frg_backward_ex failed ({rc}): {_lib.last_error()}")
                work.record_stream(torch.cuda.current_stream(dev))
        g_cv = None if out["cv"] is None else out["cv"].reshape(ctx.cv_shape)
        # settings, opts, shs, raw_opacity, raw_scale, raw_rot, means3D, shell_logits, shell_cell_verts, shell_cells, keep_mask,
        # means2D (the reference's viewspace gradient: the densification statistics read it, gaussian_model.py:404-407)
        return (None, None, out["sh"], out["op"].reshape(ctx.ro_shape), out["sc"], out["rot"],
                None if shelled else out["m3"], out["lg"], g_cv, None, None, out["m2"] if ctx.present[9] else None)


def rasterize_raw(settings, shs, raw_opacity, raw_scale, raw_rot, means3D=None, shell_logits=None, shell_cell_verts=None,
                  shell_cells=None, keep_mask=None, means2D=None, use_softmax_for_bary_coords: bool = True, modes=None):
    """-> (image [3,H,W], radii [P]).  settings: GaussianRasterizationSettings.  Exactly one of ``means3D`` [P,3]
    and (``shell_logits`` [P,6], ``shell_cell_verts`` [F,2,3,3] or [F,6,3], ``shell_cells`` [P] int64).
    raw_opacity [P] or [P,1]; raw_scale [P,3]; raw_rot [P,4]; shs [P,K,3].
    means2D (optional, [P,3] zeros with requires_grad, as the reference's callers pass it): receives the screen-space
    gradient (``viewspace_points.grad``) that densification reads.
    use_softmax_for_bary_coords = False: barycentric weights = relu(x) / sum relu(x) (frosting_model.py:716-718).
    modes: per-call forward modes {'exact_blend', 'tight_binning', 'async_sh'} overriding frg_set_option for this call."""
    if (means3D is None) == (shell_logits is None):
        raise Exception("Please provide exactly one of either means3D or the shell parameterisation!")
    opts = {"modes": modes, "use_softmax_for_bary_coords": use_softmax_for_bary_coords}
    return _RasterizeRaw.apply(settings, opts, shs, raw_opacity, raw_scale, raw_rot, means3D, shell_logits, shell_cell_verts,
                               shell_cells, keep_mask, means2D)

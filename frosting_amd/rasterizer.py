"""Host-side mirror of the reference's rasterizer Python API.

Same public names, argument meaning and error behaviour as
``diff_gaussian_rasterization/__init__.py`` of the reference
(DGR/diff_gaussian_rasterization/__init__.py:44-220), so that
``gaussian_splatting/gaussian_renderer/__init__.py:14``,
``frosting_scene/frosting_model.py:29`` and ``frosting_scene/sugar_model.py:10``
can import it unchanged (the top-level ``diff_gaussian_rasterization`` package
re-exports this module).  Underneath, the three native entry points
(``rasterize_gaussians``, ``rasterize_gaussians_backward``, ``mark_visible`` --
DGR/ext.cpp:15-18) are served by the C ABI in include/frosting_rasterizer.h;
torch is used only for device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib


def _ptr(t):
    """Device pointer of a tensor, or NULL for the reference's 'absent' encoding
    (empty tensor, DGR/diff_gaussian_rasterization/__init__.py:197-207)."""
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())


def _f32c(t, device):
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32 tensor, got {t.dtype}")  # reference: .data<float>() throws
    if t.device != device:
        raise RuntimeError(f"tensor on {t.device}, expected {device}")
    return t.contiguous()


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _Buffers:
    """The three resizable scratch tensors of the reference binding
    (rasterize_points.cu:27-33,71-78), grown through the C-ABI callbacks."""

    def __init__(self, device):
        self.device = device
        self.geom = torch.empty(0, dtype=torch.uint8, device=device)
        self.binning = torch.empty(0, dtype=torch.uint8, device=device)
        self.img = torch.empty(0, dtype=torch.uint8, device=device)

        def make(name):
            def cb(_user, nbytes):
                t = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=self.device)
                setattr(self, name, t)
                return t.data_ptr()
            return _lib.ALLOC_FN(cb)

        self.cb_geom, self.cb_binning, self.cb_img = make("geom"), make("binning"), make("img")


class _NativeOps:
    """Drop-in for the reference's pybind module ``_C`` (DGR/ext.cpp:15-18):
    identical positional signatures and return tuples (DGR/rasterize_points.h:18-67)."""

    @staticmethod
    def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                            campos, prefiltered, debug, keep_mask=None, modes=None):
        """keep_mask (extension, optional bool/uint8 [P]): Gaussians with a zero entry are left out of this
        view as if culled -- Frosting's occlusion culling without the boolean compaction of every
        per-Gaussian tensor (frosting_scene/frosting_model.py:1564-1586).
        modes (extension, optional dict): per-call forward modes {'exact_blend', 'tight_binning', 'async_sh'} that
        override the process-wide frg_set_option values for THIS call (frg_forward_args); 'forward_only': 1 = no backward
        will follow, the forward keeps nothing for one (frg_forward_args::forward_only)."""
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:57-59
        if not means3D.is_cuda:
            raise RuntimeError("frosting_amd rasterizer: means3D must live on a ROCm device (no CPU path)")
        L = _lib.lib()
        dev = means3D.device
        P, H, W = int(means3D.shape[0]), int(image_height), int(image_width)
        with torch.cuda.device(dev):
            out_color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            bufs = _Buffers(dev)
            M = int(sh.shape[1]) if sh is not None and sh.numel() != 0 else 0
            t = dict(bg=_f32c(background, dev), means=_f32c(means3D, dev), colors=_f32c(colors, dev),
                     opac=_f32c(opacity, dev), scales=_f32c(scales, dev), rots=_f32c(rotations, dev),
                     cov=_f32c(cov3D_precomp, dev), view=_f32c(viewmatrix, dev), proj=_f32c(projmatrix, dev),
                     sh=_f32c(sh, dev), campos=_f32c(campos, dev))
            if keep_mask is None and modes is None:
                rc = L.frg_forward(bufs.cb_geom, bufs.cb_binning, bufs.cb_img, None,
                                   P, int(degree), M, _ptr(t["bg"]), W, H,
                                   _ptr(t["means"]), _ptr(t["sh"]), _ptr(t["colors"]), _ptr(t["opac"]),
                                   _ptr(t["scales"]), float(scale_modifier), _ptr(t["rots"]), _ptr(t["cov"]),
                                   _ptr(t["view"]), _ptr(t["proj"]), _ptr(t["campos"]),
                                   float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                                   _ptr(out_color), _ptr(radii) if P else None, int(bool(debug)), _stream_ptr(dev))
            else:
                if keep_mask is not None and (keep_mask.shape != (P,) or keep_mask.device != dev or
                                              keep_mask.dtype not in (torch.bool, torch.uint8)):
                    raise RuntimeError("keep_mask must be a bool / uint8 tensor of shape (num_points,) on the Gaussians' device")
                mask = None if keep_mask is None else keep_mask.contiguous()

                def vp(x):
                    return None if x is None else x.value
                a = _lib.ForwardArgs(
                    struct_size=C.sizeof(_lib.ForwardArgs), geometry_alloc=bufs.cb_geom, binning_alloc=bufs.cb_binning,
                    image_alloc=bufs.cb_img, user=None, P=P, D=int(degree), M=M, background=vp(_ptr(t["bg"])), width=W,
                    height=H, means3D=vp(_ptr(t["means"])), shs=vp(_ptr(t["sh"])), colors_precomp=vp(_ptr(t["colors"])),
                    opacities=vp(_ptr(t["opac"])), scales=vp(_ptr(t["scales"])), scale_modifier=float(scale_modifier),
                    rotations=vp(_ptr(t["rots"])), cov3D_precomp=vp(_ptr(t["cov"])), viewmatrix=vp(_ptr(t["view"])),
                    projmatrix=vp(_ptr(t["proj"])), cam_pos=vp(_ptr(t["campos"])), tan_fovx=float(tan_fovx),
                    tan_fovy=float(tan_fovy), prefiltered=int(bool(prefiltered)), out_color=out_color.data_ptr(),
                    radii=radii.data_ptr() if P else None, debug=int(bool(debug)), hip_stream=_stream_ptr(dev).value,
                    instance_capacity=0, keep_mask=mask.data_ptr() if (P and mask is not None) else None,
                    **_lib.mode_fields(modes))
                rc = L.frg_forward_ex(C.byref(a))
        if rc < 0:
            raise RuntimeError(f"frg_forward failed ({rc}): {_lib.last_error()}")
        return rc, out_color, radii, bufs.geom, bufs.binning, bufs.img

    @staticmethod
    def rasterize_gaussians_masked(*args):
        """(the 19 arguments of rasterize_gaussians, keep_mask) -- same name as the compiled module's export."""
        return _NativeOps.rasterize_gaussians(*args[:19], keep_mask=args[19])

    @staticmethod
    def rasterize_gaussians_forward_only(*args):
        """(the 19 arguments of rasterize_gaussians, keep_mask or an empty tensor): the forward of a call no backward will
        follow (frg_forward_args::forward_only) -- same name as the compiled module's export."""
        mask = args[19] if len(args) > 19 and args[19] is not None and args[19].numel() else None
        return _NativeOps.rasterize_gaussians(*args[:19], keep_mask=mask, modes={"forward_only": 1})

    @staticmethod
    def rasterize_gaussians_ex(*args):
        """(the 19 arguments of rasterize_gaussians, keep_mask or an empty tensor, exact_blend -1 | 0 | 1, forward_only):
        every forward option of the Python layer in one export -- same name as the compiled module's.  exact_blend -1 =
        the process-wide option, 0 | 1 = the blend arithmetic of THIS call."""
        mask = args[19] if args[19] is not None and args[19].numel() else None
        modes = {"forward_only": int(bool(args[21]))}
        if int(args[20]) >= 0:
            modes["exact_blend"] = int(args[20])
        return _NativeOps.rasterize_gaussians(*args[:19], keep_mask=mask, modes=modes)

    @staticmethod
    def get_option(name):
        return _lib.get_option(name)

    @staticmethod
    def rasterize_gaussians_backward_ex(*args):
        """(the 21 arguments of rasterize_gaussians_backward, exact_blend -1 | 0 | 1): the forward's blend arithmetic handed
        over by the caller, who carried it beside the buffers (the autograd ctx) -- same name as the compiled module's export."""
        return _NativeOps.rasterize_gaussians_backward(*args[:21], exact_blend=int(args[21]))

    @staticmethod
    def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh,
                                     degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug, exact_blend=-1):
        """exact_blend (extension): -1 = the arithmetic of the forward that filled the buffers (what the library remembers
        of it, else what that forward stamped into imageBuffer); 0 | 1 = stated by the caller (frg_backward_args::exact_blend)."""
        L = _lib.lib()
        dev = means3D.device
        P = int(means3D.shape[0])
        H, W = int(dL_dout_color.shape[1]), int(dL_dout_color.shape[2])  # rasterize_points.cu:142-143
        M = int(sh.shape[1]) if sh is not None and sh.numel() != 0 else 0
        with torch.cuda.device(dev):
            def e(*shape):
                return torch.empty(shape, dtype=torch.float32, device=dev)
            has_sh = sh is not None and sh.numel() != 0
            has_sr = scales is not None and scales.numel() != 0
            dL_dmeans3D, dL_dmeans2D, dL_dcolors = e(P, 3), e(P, 3), e(P, 3)
            dL_dopacity, dL_dcov3D = e(P, 1), e(P, 6)   # dL_dconic is an intermediate the reference never returns (:195)
            # rows the kernels do not write (absent input) stay zero, as in the reference's zero-allocated outputs
            dL_dsh = e(P, M, 3) if has_sh else torch.zeros((P, M, 3), dtype=torch.float32, device=dev)
            dL_dscales = e(P, 3) if has_sr else torch.zeros((P, 3), dtype=torch.float32, device=dev)
            dL_drotations = e(P, 4) if has_sr else torch.zeros((P, 4), dtype=torch.float32, device=dev)
            if P != 0:
                ws_bytes = int(L.frg_backward_workspace_bytes(P, int(R)))
                workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                t = dict(bg=_f32c(background, dev), means=_f32c(means3D, dev), colors=_f32c(colors, dev),
                         scales=_f32c(scales, dev), rots=_f32c(rotations, dev), cov=_f32c(cov3D_precomp, dev),
                         view=_f32c(viewmatrix, dev), proj=_f32c(projmatrix, dev), sh=_f32c(sh, dev),
                         campos=_f32c(campos, dev), dpix=_f32c(dL_dout_color, dev), radii=radii.contiguous())
                if int(exact_blend) < 0:      # the reference-shaped entry point (rasterizer.h:58-84)
                    rc = L.frg_backward(P, int(degree), M, int(R), _ptr(t["bg"]), W, H,
                                        _ptr(t["means"]), _ptr(t["sh"]), _ptr(t["colors"]),
                                        _ptr(t["scales"]), float(scale_modifier), _ptr(t["rots"]), _ptr(t["cov"]),
                                        _ptr(t["view"]), _ptr(t["proj"]), _ptr(t["campos"]),
                                        float(tan_fovx), float(tan_fovy), _ptr(t["radii"]),
                                        _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(t["dpix"]),
                                        _ptr(dL_dmeans2D), None, _ptr(dL_dopacity), _ptr(dL_dcolors),
                                        _ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dsh) if has_sh else None,
                                        _ptr(dL_dscales) if has_sr else None, _ptr(dL_drotations) if has_sr else None,
                                        _ptr(workspace), ws_bytes, int(bool(debug)), _stream_ptr(dev))
                else:
                    def vp(x):
                        return None if x is None else x.value
                    a = _lib.BackwardArgs(
                        struct_size=C.sizeof(_lib.BackwardArgs), P=P, D=int(degree), M=M, R=int(R), background=vp(_ptr(t["bg"])),
                        width=W, height=H, means3D=vp(_ptr(t["means"])), shs=vp(_ptr(t["sh"])), colors_precomp=vp(_ptr(t["colors"])),
                        scales=vp(_ptr(t["scales"])), scale_modifier=float(scale_modifier), rotations=vp(_ptr(t["rots"])),
                        cov3D_precomp=vp(_ptr(t["cov"])), viewmatrix=vp(_ptr(t["view"])), projmatrix=vp(_ptr(t["proj"])),
                        campos=vp(_ptr(t["campos"])), tan_fovx=float(tan_fovx), tan_fovy=float(tan_fovy), radii=vp(_ptr(t["radii"])),
                        geom_buffer=vp(_ptr(geomBuffer)), binning_buffer=vp(_ptr(binningBuffer)), image_buffer=vp(_ptr(imageBuffer)),
                        dL_dpix=vp(_ptr(t["dpix"])), dL_dmean2D=vp(_ptr(dL_dmeans2D)), dL_dconic=None, dL_dopacity=vp(_ptr(dL_dopacity)),
                        dL_dcolor=vp(_ptr(dL_dcolors)), dL_dmean3D=vp(_ptr(dL_dmeans3D)), dL_dcov3D=vp(_ptr(dL_dcov3D)),
                        dL_dsh=vp(_ptr(dL_dsh)) if has_sh else None, dL_dscale=vp(_ptr(dL_dscales)) if has_sr else None,
                        dL_drot=vp(_ptr(dL_drotations)) if has_sr else None, workspace=vp(_ptr(workspace)), workspace_bytes=ws_bytes,
                        debug=int(bool(debug)), hip_stream=_stream_ptr(dev).value, exact_blend=int(exact_blend) + 1)
                    rc = L.frg_backward_ex(C.byref(a))
                if rc < 0:
                    raise RuntimeError(f"frg_backward failed ({rc}): {_lib.last_error()}")
                # keep the workspace alive until the stream has consumed it
                workspace.record_stream(torch.cuda.current_stream(dev))
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        L = _lib.lib()
        dev = means3D.device
        P = int(means3D.shape[0])
        present = torch.zeros((P,), dtype=torch.bool, device=dev)
        if P != 0:
            with torch.cuda.device(dev):
                m, v, p = _f32c(means3D, dev), _f32c(viewmatrix, dev), _f32c(projmatrix, dev)
                rc = L.frg_mark_visible(P, _ptr(m), _ptr(v), _ptr(p), _ptr(present), _stream_ptr(dev))
            if rc < 0:
                raise RuntimeError(f"frg_mark_visible failed ({rc}): {_lib.last_error()}")
        return present


_C = _NativeOps()


class GaussianRasterizationSettings(NamedTuple):
    """The 12 per-view settings, same fields and order as the reference
    (DGR/diff_gaussian_rasterization/__init__.py:157-169)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _snapshot(args):
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _make_autograd_function(ops):
    """The reference's _RasterizeGaussians (__init__.py:44-155) over one native binding `ops`
    (an object exporting rasterize_gaussians / rasterize_gaussians_masked / rasterize_gaussians_backward):
    the ctypes binding above, or the compiled torch extension diff_gaussian_rasterization._C."""

    class _RasterizeGaussians(torch.autograd.Function):
        """Autograd wiring.  Gradients come back in input order; the settings argument gets None."""

        @staticmethod
        def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                    raster_settings, keep_mask=None):
            s = raster_settings
            native_args = (s.bg, means3D, colors_precomp, opacities, scales, rotations, s.scale_modifier, cov3Ds_precomp,
                           s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.image_height, s.image_width, sh,
                           s.sh_degree, s.campos, s.prefiltered, s.debug)

            # no input needs a gradient (or torch.no_grad()): no backward can follow -- the native forward keeps nothing for one
            forward_only = not any(ctx.needs_input_grad) and hasattr(ops, "rasterize_gaussians_forward_only")
            # The modes of THIS forward travel with the autograd ctx, beside the buffers (the reference's ctx carries
            # num_rendered the same way, __init__.py:93-97): the blend arithmetic is fixed here -- the process option read once
            # and handed to the native forward as its per-call mode -- and given back to the native backward, which then
            # depends neither on what the library remembers of this forward nor on the option's value by then.
            carried = hasattr(ops, "rasterize_gaussians_ex") and hasattr(ops, "rasterize_gaussians_backward_ex")
            exact = int(ops.get_option("exact_blend")) if carried else -1

            def run():
                if carried:
                    return ops.rasterize_gaussians_ex(*native_args, keep_mask if keep_mask is not None else torch.empty(0),
                                                      exact, forward_only)
                if forward_only:
                    return ops.rasterize_gaussians_forward_only(*native_args, keep_mask if keep_mask is not None else torch.empty(0))
                if keep_mask is None:
                    return ops.rasterize_gaussians(*native_args)
                return ops.rasterize_gaussians_masked(*native_args, keep_mask)
            if s.debug:
                saved = _snapshot(native_args)
                try:
                    out = run()
                except Exception:
                    torch.save(saved, "snapshot_fw.dump")  # same replay fixture as the reference (:83-90)
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                    raise
            else:
                out = run()
            num_rendered, color, radii, geom, binning, img = out
            ctx.raster_settings = s
            ctx.num_rendered = num_rendered
            ctx.exact_blend, ctx.forward_only = exact, forward_only
            ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
            ctx.mark_non_differentiable(radii)
            return color, radii

        @staticmethod
        def backward(ctx, grad_out_color, _grad_radii):
            s = ctx.raster_settings
            colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
            native_args = (s.bg, means3D, radii, colors_precomp, scales, rotations, s.scale_modifier, cov3Ds_precomp,
                           s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, grad_out_color, sh, s.sh_degree, s.campos,
                           geom, ctx.num_rendered, binning, img, s.debug)
            if ctx.forward_only:
                raise RuntimeError("backward of a forward that was run with no input requiring a gradient (forward_only): "
                                   "nothing was kept for it")
            backward_op = ops.rasterize_gaussians_backward
            if ctx.exact_blend >= 0:          # the forward's arithmetic, carried by this ctx
                def backward_op(*a):
                    return ops.rasterize_gaussians_backward_ex(*a, ctx.exact_blend)
            if s.debug:
                saved = _snapshot(native_args)
                try:
                    grads = backward_op(*native_args)
                except Exception:
                    torch.save(saved, "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                    raise
            else:
                grads = backward_op(*native_args)
            g_means2D, g_colors, g_opac, g_means3D, g_cov3D, g_sh, g_scales, g_rots = grads
            return g_means3D, g_means2D, g_sh, g_colors, g_opac, g_scales, g_rots, g_cov3D, None, None

    return _RasterizeGaussians


_RasterizeGaussians = _make_autograd_function(_C)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, keep_mask=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, keep_mask)


class GaussianRasterizer(nn.Module):
    """``GaussianRasterizer(raster_settings)(means3D=..., means2D=..., opacities=..., shs=|colors_precomp=,
    scales=+rotations=|cov3D_precomp=) -> (image [3,H,W], radii [P] int32)`` (reference :171-220)."""

    _ops = _C                              # native binding (subclasses rebind: make_rasterizer_class)
    _function = _RasterizeGaussians

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            s = self.raster_settings
            return self._ops.mark_visible(positions, s.viewmatrix, s.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, keep_mask=None):
        """keep_mask: extension over the reference signature (optional bool [P]); see
        _NativeOps.rasterize_gaussians."""
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        have_sr = scales is not None and rotations is not None
        partial_sr = (scales is not None) or (rotations is not None)
        if (not have_sr and cov3D_precomp is None) or (partial_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])  # the reference's encoding of an absent input
        return self._function.apply(
            means3D, means2D,
            empty if shs is None else shs,
            empty if colors_precomp is None else colors_precomp,
            opacities,
            empty if scales is None else scales,
            empty if rotations is None else rotations,
            empty if cov3D_precomp is None else cov3D_precomp,
            self.raster_settings, keep_mask)


def make_rasterizer_class(ops):
    """GaussianRasterizer bound to another native binding with the same exports (the compiled
    torch extension): returns (GaussianRasterizer subclass, its autograd Function)."""
    fn = _make_autograd_function(ops)
    cls = type("GaussianRasterizer", (GaussianRasterizer,), {"_ops": ops, "_function": fn, "__doc__": GaussianRasterizer.__doc__})
    return cls, fn

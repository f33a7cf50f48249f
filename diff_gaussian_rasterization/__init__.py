"""Drop-in import name for the reference's rasterizer package.

``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer``
is hard-coded at the reference's three call sites
(gaussian_splatting/gaussian_renderer/__init__.py:14, frosting_scene/frosting_model.py:29,
frosting_scene/sugar_model.py:10).  As in the reference (``from . import _C``,
DGR/diff_gaussian_rasterization/__init__.py:15) the native entry points come from the compiled
torch extension ``diff_gaussian_rasterization._C`` (setup.py; frosting_amd/csrc/torch_ext/torch_binding.cpp
over the gfx950 HIP library); the Python above it -- settings tuple, argument rules, autograd wiring --
is frosting_amd.rasterizer.  No fallback: without the built extension the import fails.
"""
import torch  # noqa: F401  (loads libtorch / libc10_hip, which _C links against)

try:
    from . import _C
except ImportError as e:  # pragma: no cover - build problem, reported loudly
    raise ImportError(
        "diff_gaussian_rasterization._C is not built: run `python setup.py build_ext --inplace` (or "
        "`python -c 'import __graft_entry__ as g; g.build()'`) in the repository root.  There is no fallback "
        f"path.  ({e})") from e

from frosting_amd.rasterizer import GaussianRasterizationSettings, make_rasterizer_class

GaussianRasterizer, _RasterizeGaussians = make_rasterizer_class(_C)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, keep_mask=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, keep_mask)


__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]

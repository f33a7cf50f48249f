"""Drop-in import name for the reference's rasterizer package.

``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer``
is hard-coded at the reference's three call sites
(gaussian_splatting/gaussian_renderer/__init__.py:14, frosting_scene/frosting_model.py:29,
frosting_scene/sugar_model.py:10).  Everything is served by frosting_amd (gfx950 HIP).
"""
from frosting_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    _C,
    _RasterizeGaussians,
    rasterize_gaussians,
)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]

"""frosting_amd -- MI355X-native differentiable Gaussian-splat rasterizer.

Only the hot path named by BASELINE.json's north_star lives here: the
``diff_gaussian_rasterization`` operator (forward + backward) as hand-written
gfx950 HIP kernels behind a C ABI (include/frosting_rasterizer.h), its Python
host mirror (GaussianRasterizer / GaussianRasterizationSettings), the SH colour
helper, the triangle occlusion raster and the view-parallel gradient exchange.
"""
__version__ = "0.1.0"

"""Real spherical-harmonics colour helpers (degrees 0-4) in plain torch.

Host-side mirror of the reference's ``eval_sh`` / ``RGB2SH`` / ``SH2RGB``
(frosting_utils/spherical_harmonics.py:117-178; same basis, sign and ordering
conventions as the in-kernel evaluation at DGR/cuda_rasterizer/forward.cu:20-71).
Formulated as basis(dirs) . coefficients so that the basis can be reused; the
rasterizer kernels evaluate the same polynomials on the device.
"""
from __future__ import annotations

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)
C4 = (2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892, 0.10578554691520431,
      -0.6690465435572892, 0.47308734787878004, -1.7701307697799304, 0.6258357354491761)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """[..., 3] unit directions -> [..., (deg+1)^2] basis values (signs folded in)."""
    if not 0 <= deg <= 4:
        raise ValueError("SH degree must be in 0..4")
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    cols = [torch.full_like(x, C0)]
    if deg >= 1:
        cols += [-C1 * y, C1 * z, -C1 * x]
    if deg >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        cols += [C2[0] * xy, C2[1] * yz, C2[2] * (2.0 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg >= 3:
        cols += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
                 C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
                 C3[6] * x * (xx - 3 * yy)]
    if deg >= 4:
        cols += [C4[0] * xy * (xx - yy), C4[1] * yz * (3 * xx - yy), C4[2] * xy * (7 * zz - 1),
                 C4[3] * yz * (7 * zz - 3), C4[4] * (zz * (35 * zz - 30) + 3), C4[5] * xz * (7 * zz - 3),
                 C4[6] * (xx - yy) * (7 * zz - 1), C4[7] * xz * (xx - 3 * yy),
                 C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(cols, dim=-1)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh: [..., C, K] coefficients (channel-major, as the reference's callers pass
    them -- frosting_model.py:1348), dirs: [..., 3] unit vectors -> [..., C]."""
    k = (deg + 1) ** 2
    if sh.shape[-1] < k:
        raise ValueError(f"degree {deg} needs {k} coefficients, got {sh.shape[-1]}")
    basis = sh_basis(deg, dirs)
    return (sh[..., :k] * basis.unsqueeze(-2)).sum(dim=-1)


def rgb_to_sh(rgb):
    return (rgb - 0.5) / C0


def sh_to_rgb(sh):
    return sh * C0 + 0.5


RGB2SH, SH2RGB = rgb_to_sh, sh_to_rgb


def points_rgb(means3D, shs, campos, deg):
    """View-dependent colour exactly as the rasterizer derives it
    (Frosting.get_points_rgb, frosting_model.py:1304-1352): normalised
    direction from the camera, SH evaluation, +0.5, clamp at 0.  shs is [P,K,3]."""
    d = means3D - campos.reshape(1, 3)
    d = d / d.norm(dim=-1, keepdim=True)
    col = eval_sh(deg, shs.transpose(-1, -2), d) + 0.5
    return col.clamp_min(0.0)

"""Independent float64 autograd formulation of the splatting pipeline (tiny scenes).

Not a restatement of the reference's code: the EWA projection is written as
matrix algebra (Sigma' = J W Sigma W^T J^T), compositing as an exclusive cumprod
over depth-ordered Gaussians per pixel, and gradients come from torch.autograd.
The discrete decisions (cull, radius/tile rectangle, alpha thresholds, early
termination) are taken from detached values, exactly the non-differentiable
choices the rasterizer makes.  Used to pin the analytic backward of the C oracle
(and through it the HIP kernels) against an independent derivation.
"""
from __future__ import annotations

import math

import torch

from frosting_amd.sh import sh_basis


def render(means3D, scales, rotations, opacities, shs, cam, bg, sh_degree):
    """All inputs float64 tensors (requires_grad as desired).  Returns image [3,H,W]."""
    dt = torch.float64
    W, H = cam.image_width, cam.image_height
    vm = cam.viewmatrix.to(dt)      # row-vector convention: p_row @ vm
    pm = cam.projmatrix.to(dt)
    campos = cam.campos.to(dt)
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    t = (ph @ vm)[:, :3]
    hom = ph @ pm
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    visible = t[:, 2].detach() > 0.2

    # covariance
    q = rotations
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(P, 3, 3)
    Sig = R @ torch.diag_embed(scales * scales) @ R.transpose(1, 2)
    fx, fy = W / (2 * cam.tanfovx), H / (2 * cam.tanfovy)
    limx, limy = 1.3 * cam.tanfovx, 1.3 * cam.tanfovy
    tz = t[:, 2]
    # outside 1.3x the field of view the reference freezes the clamped coordinate
    # (x_grad_mul / y_grad_mul = 0 and no dependence on t.z, backward.cu:175-176,262-264)
    rx, ry = (t[:, 0] / tz).detach(), (t[:, 1] / tz).detach()
    tx = torch.where(rx.abs() <= limx, t[:, 0], (torch.clamp(rx, -limx, limx) * tz).detach())
    ty = torch.where(ry.abs() <= limy, t[:, 1], (torch.clamp(ry, -limy, limy) * tz).detach())
    J = torch.zeros(P, 2, 3, dtype=dt)
    J[:, 0, 0] = fx / tz
    J[:, 0, 2] = -fx * tx / (tz * tz)
    J[:, 1, 1] = fy / tz
    J[:, 1, 2] = -fy * ty / (tz * tz)
    Wr = vm[:3, :3].t()             # world -> camera rotation (column-vector form)
    Tm = J @ Wr
    cov2 = Tm @ Sig @ Tm.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    conA, conB, conC = c / det, -b / det, a / det
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()

    # colour
    d = means3D - campos[None]
    d = d / d.norm(dim=1, keepdim=True)
    basis = sh_basis(sh_degree, d)                         # [P,K]
    K = basis.shape[1]
    col = (basis[:, :, None] * shs[:, :K, :]).sum(1) + 0.5
    col = torch.clamp_min(col, 0.0)

    # tile rectangles (detached integers)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pxd, pyd = px.detach(), py.detach()
    x0 = torch.clamp(torch.trunc((pxd - radius) / 16), 0, gx)
    y0 = torch.clamp(torch.trunc((pyd - radius) / 16), 0, gy)
    x1 = torch.clamp(torch.trunc((pxd + radius + 15) / 16), 0, gx)
    y1 = torch.clamp(torch.trunc((pyd + radius + 15) / 16), 0, gy)
    visible = visible & ((x1 - x0) * (y1 - y0) > 0) & (det.detach() != 0)

    order = torch.argsort(t[:, 2].detach().float(), stable=True)  # float32 depth keys, ties by index
    order = order[visible[order]]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    img = torch.zeros(3, H, W, dtype=dt)
    Tacc = torch.ones(H, W, dtype=dt)
    alive = torch.ones(H, W, dtype=torch.bool)
    tile_x, tile_y = torch.floor(xs / 16), torch.floor(ys / 16)
    for i in order.tolist():
        in_rect = (tile_x >= x0[i]) & (tile_x < x1[i]) & (tile_y >= y0[i]) & (tile_y < y1[i])
        dx, dy = px[i] - xs, py[i] - ys
        power = -0.5 * (conA[i] * dx * dx + conC[i] * dy * dy) - conB[i] * dx * dy
        alpha = torch.clamp(opacities[i, 0] * torch.exp(power), max=0.99)
        ok = in_rect & alive & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
        test_T = Tacc * (1 - alpha)
        stop = ok & (test_T.detach() < 1e-4)
        alive = alive & ~stop
        use = ok & ~stop
        w = torch.where(use, alpha * Tacc, torch.zeros_like(alpha))
        img = img + col[i][:, None, None] * w[None]
        Tacc = torch.where(use, test_T, Tacc)
    return img + Tacc[None] * bg.to(dt)[:, None, None]


# ---- photometric loss restatement (frosting_utils/loss_utils.py:17-62, refine.py:407-409) ---------
def photometric_loss_ref(pred, gt, lambda_dssim=0.2, window_size=11, sigma=1.5):
    """(1 - l) * l1_loss + l * (1 - ssim), every step as the reference writes it (grouped conv2d with
    the 2-D outer-product window, padding = window_size // 2).  pred, gt: [C,H,W]."""
    import torch.nn.functional as F
    from math import exp
    g1 = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    g1 = (g1 / g1.sum()).unsqueeze(1)
    C = pred.shape[-3]
    window = g1.mm(g1.t()).float().unsqueeze(0).unsqueeze(0).expand(C, 1, window_size, window_size).contiguous()
    window = window.to(device=pred.device, dtype=pred.dtype)
    a, b = pred.unsqueeze(0), gt.unsqueeze(0)
    pad = window_size // 2
    mu1 = F.conv2d(a, window, padding=pad, groups=C)
    mu2 = F.conv2d(b, window, padding=pad, groups=C)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(a * a, window, padding=pad, groups=C) - mu1_sq
    sigma2_sq = F.conv2d(b * b, window, padding=pad, groups=C) - mu2_sq
    sigma12 = F.conv2d(a * b, window, padding=pad, groups=C) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return (1.0 - lambda_dssim) * torch.abs(pred - gt).mean() + lambda_dssim * (1.0 - ssim_map.mean())

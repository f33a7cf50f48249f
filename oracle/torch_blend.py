"""TEST INFRASTRUCTURE / CPU BASELINE: the front-to-back alpha blend of the reference (forward.cu:261-374) and its
gradients in PURE PyTorch on the CPU -- the "pure-PyTorch CPU alpha-blend baseline" BASELINE.json's north_star asks
bench.py to time on the host cores.  Never imported by the product path.

Not a restatement of the reference's kernel: per tile, the list entries are an [n, 256] tensor against the tile's pixels,
the transmittance is an exclusive cumulative product along the list, the reference's three per-pixel tests (power > 0,
alpha < 1/255, T (1 - alpha) < 1e-4 -> stop) are masks, and the gradients come from torch.autograd on per-tile leaves.
The list is walked in chunks so that a saturated tile stops early, as any sensible torch implementation would.
Checked against the C oracle in tests/test_oracle_cpu.py."""
from __future__ import annotations

import time

import torch

TILE = 16


def _tile_pixels(tx, ty, W, H):
    ys, xs = torch.meshgrid(torch.arange(ty * TILE, ty * TILE + TILE), torch.arange(tx * TILE, tx * TILE + TILE), indexing="ij")
    inside = (xs < W) & (ys < H)
    return xs.reshape(-1).float(), ys.reshape(-1).float(), inside.reshape(-1)


def blend_tile(xy, co, col, pxf, pyf, inside, bg, chunk=512):
    """xy [n,2], co [n,4] (conic a, b, c, opacity), col [n,3] in list order; pixels [256].  Returns (C [3,256],
    final_T [256], n_contrib [256]) with the reference's semantics; differentiable in xy / co / col."""
    n = xy.shape[0]
    T = torch.ones_like(pxf)
    alive = inside.clone()
    C = torch.zeros(3, pxf.shape[0], dtype=xy.dtype)
    last = torch.zeros(pxf.shape[0], dtype=torch.long)
    for s in range(0, n, chunk):
        if not bool(alive.any()):
            break
        e = min(n, s + chunk)
        dx = xy[s:e, 0:1] - pxf[None, :]
        dy = xy[s:e, 1:2] - pyf[None, :]
        a, b, c, o = co[s:e, 0:1], co[s:e, 1:2], co[s:e, 2:3], co[s:e, 3:4]
        power = -0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy
        alpha = torch.clamp(o * torch.exp(power), max=0.99)
        keep = (power <= 0) & (alpha >= 1.0 / 255.0) & alive[None, :]
        alpha = torch.where(keep, alpha, torch.zeros_like(alpha))
        incl = torch.cumprod(1.0 - alpha, 0) * T[None, :]          # transmittance behind entry i
        # the reference stops a pixel at the first kept entry whose test_T = T (1 - alpha) drops below 1e-4; that
        # entry and everything after it contribute nothing
        stop = keep & (incl.detach() < 1e-4)
        dead = torch.cumsum(stop.long(), 0) > 0
        w_keep = keep & ~dead
        alpha = torch.where(w_keep, alpha, torch.zeros_like(alpha))
        incl = torch.cumprod(1.0 - alpha, 0) * T[None, :]
        excl = torch.cat([T[None, :], incl[:-1]], 0)
        w = alpha * excl
        C = C + torch.einsum("np,nc->cp", w, col[s:e])
        T = incl[-1]
        pos = torch.arange(s + 1, e + 1)[:, None].expand(-1, pxf.shape[0])
        last = torch.maximum(last, torch.where(w_keep, pos, torch.zeros_like(pos)).max(0).values)
        alive = alive & ~dead[-1]
    out = C + bg[:, None] * T[None, :]
    return out, T, last


def render(means2D, conic_opacity, colors, ranges, point_list, bg, W, H, tiles=None, dL_dimage=None, chunk=512):
    """Blend the given tiles (default: all).  Returns dict(image [3,H,W], final_T, n_contrib, seconds) and, with
    dL_dimage [3,H,W], the gradients dL_dmeans2D [P,2], dL_dconic_opacity [P,4], dL_dcolors [P,3] (autograd)."""
    t0 = time.perf_counter()
    gx = (W + TILE - 1) // TILE
    P = means2D.shape[0]
    image = torch.zeros(3, H, W)
    fT = torch.ones(H, W)
    ncon = torch.zeros(H, W, dtype=torch.long)
    grads = None
    if dL_dimage is not None:
        grads = dict(dL_dmeans2D=torch.zeros(P, 2), dL_dconic_opacity=torch.zeros(P, 4), dL_dcolors=torch.zeros(P, 3))
    pl = point_list.long()
    for t in (range(ranges.shape[0]) if tiles is None else tiles):
        tx, ty = t % gx, t // gx
        r0, r1 = int(ranges[t, 0]), int(ranges[t, 1])
        pxf, pyf, inside = _tile_pixels(tx, ty, W, H)
        idx = pl[r0:r1]
        xy, co, col = means2D[idx], conic_opacity[idx], colors[idx]
        if grads is not None:
            xy, co, col = (v.detach().clone().requires_grad_(True) for v in (xy, co, col))
        out, T, last = blend_tile(xy, co, col, pxf, pyf, inside, bg, chunk)
        y0, x0 = ty * TILE, tx * TILE
        hh, ww = min(TILE, H - y0), min(TILE, W - x0)
        image[:, y0:y0 + hh, x0:x0 + ww] = out.detach().view(3, TILE, TILE)[:, :hh, :ww]
        fT[y0:y0 + hh, x0:x0 + ww] = T.detach().view(TILE, TILE)[:hh, :ww]
        ncon[y0:y0 + hh, x0:x0 + ww] = last.view(TILE, TILE)[:hh, :ww]
        if grads is not None and r1 > r0:
            g = torch.zeros(3, TILE, TILE)
            g[:, :hh, :ww] = dL_dimage[:, y0:y0 + hh, x0:x0 + ww]
            out.backward(g.view(3, -1))
            grads["dL_dmeans2D"].index_add_(0, idx, xy.grad)
            grads["dL_dconic_opacity"].index_add_(0, idx, co.grad)
            grads["dL_dcolors"].index_add_(0, idx, col.grad)
    res = dict(image=image, final_T=fT, n_contrib=ncon, seconds=time.perf_counter() - t0)
    if grads is not None:
        res.update(grads)
    return res

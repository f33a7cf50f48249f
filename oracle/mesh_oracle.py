"""TEST INFRASTRUCTURE: CPU z-buffer restatement of the triangle occlusion raster.

The reference's own rasterizer for this side op is nvdiffrast (third-party,
`git clone` of HEAD at install.py:33-36, not pinned and not in /root/reference),
called at frosting_utils/nvdiffrast.py:53.  With neither its source nor any
reference test fixture available, PARITY FOR THIS SIDE OP IS UNPINNED: this file
restates nvdiffrast's published contract (rast = (u, v, z/w, triangle_id+1);
perspective-correct barycentrics; pixel centres at (i+0.5)/W*2-1; depth clip
-1 <= z/w <= 1; nearest fragment wins) as a brute-force float64 evaluation, and the
acceptance criterion is the one SURVEY.md 8(c) sets: identical visible-face sets up
to faces covering less than a pixel.  Pure numpy, tiny meshes only.
"""
from __future__ import annotations

import numpy as np


def rasterize(pos, tri, height, width):
    """pos [V,4] clip space, tri [F,3] -> rast [H,W,4] float64 (u, v, z/w, id+1)."""
    pos = np.asarray(pos, dtype=np.float64)
    tri = np.asarray(tri, dtype=np.int64)
    H, W = height, width
    X = (np.arange(W) + 0.5) / W * 2 - 1
    Y = (np.arange(H) + 0.5) / H * 2 - 1
    XX, YY = np.meshgrid(X, Y)                       # [H,W]
    best = np.full((H, W), np.inf)
    rast = np.zeros((H, W, 4))
    for f in range(tri.shape[0]):
        v = pos[tri[f]]                               # [3,4]
        Mt = np.stack([v[:, 0], v[:, 1], v[:, 3]])    # rows x, y, w; columns = vertices
        det = np.linalg.det(Mt)
        if det == 0 or not np.isfinite(det):
            continue
        inv = np.linalg.inv(Mt)                        # rows: edge functions (a, b, c)
        e = inv[:, 0, None, None] * XX + inv[:, 1, None, None] * YY + inv[:, 2, None, None]   # [3,H,W]
        s = e.sum(0)
        inside = (e >= 0).all(0) & (s > 0)
        if not inside.any():
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            b = e / s
            zc = (b * v[:, 2, None, None]).sum(0)
            wc = (b * v[:, 3, None, None]).sum(0)
            zw = zc / wc
        ok = inside & (zw >= -1) & (zw <= 1) & (zw < best)   # strict <: smaller id wins ties
        best = np.where(ok, zw, best)
        rast[ok, 0] = b[0][ok]; rast[ok, 1] = b[1][ok]; rast[ok, 2] = zw[ok]; rast[ok, 3] = f + 1
    return rast

"""TEST INFRASTRUCTURE: CPU z-buffer restatement of the triangle occlusion raster.

The reference's own rasterizer for this side op is nvdiffrast (third-party,
`git clone` of HEAD at install.py:33-36, not pinned and not in /root/reference),
called at frosting_utils/nvdiffrast.py:53.  With neither its source nor any
reference test fixture available, PARITY FOR THIS SIDE OP IS UNPINNED against nvdiffrast
itself.  What IS specified is pinned here and tested (tests/test_mesh_oracle_cpu.py,
tests/test_gpu_mesh.py):

  * output contract of `dr.rasterize` as the reference consumes it
    (frosting_utils/nvdiffrast.py:53-58): rast = (u, v, z/w, triangle_id + 1), zeros where empty;
    u, v = perspective-correct barycentrics of vertices 0 and 1; row j <-> NDC y increasing;
  * sample positions: pixel centres ((i + 0.5) / W * 2 - 1, (j + 0.5) / H * 2 - 1) (OpenGL);
  * depth: fragments with z/w outside [-1, 1] are clipped; the nearest z/w wins, equal depths go to
    the smaller triangle id; no face culling;
  * fill rule: top-left on the edge normal -- a centre exactly on an edge belongs to the triangle for
    which that edge's function e = aX + bY + c has a > 0, or a == 0 and b > 0 -- with the edge
    function of a shared edge computed identically (up to exact negation) from both sides, so a
    closed mesh is covered without double hits and without cracks.

Float64 numpy.  `rasterize` is the brute-force form (every triangle against every pixel, tiny
meshes); `rasterize_windowed` evaluates each triangle only over its bounding box and scales to the
C4 shell (200 k triangles at 1600x1056) -- the two are checked against each other.
Acceptance for the HIP kernel: identical pix_to_face up to faces covering < 1 px.
"""
from __future__ import annotations

import numpy as np


def _edge_coeffs(p, q, sgn):
    """Coefficients of sgn * det[(X,Y,1), p, q] built from the end points in canonical (position) order.
    p, q: [...,3] arrays of (x, y, w) that hold float32 values; returns a, b, c float64."""
    swap = (q[..., 0] < p[..., 0]) | ((q[..., 0] == p[..., 0]) & ((q[..., 1] < p[..., 1]) |
                                                                ((q[..., 1] == p[..., 1]) & (q[..., 2] < p[..., 2]))))
    lo = np.where(swap[..., None], q, p)
    hi = np.where(swap[..., None], p, q)
    lx, ly, lw, hx, hy, hw = lo[..., 0], lo[..., 1], lo[..., 2], hi[..., 0], hi[..., 1], hi[..., 2]
    # products of float32-valued doubles are exact, so each difference is rounded once (== the kernel's fma form)
    ca, cb, cc = ly * hw - hy * lw, hx * lw - lx * hw, lx * hy - hx * ly
    s = np.where(swap, -sgn, sgn)
    return s * ca, s * cb, s * cc


def _setup(pos, tri):
    pos = np.asarray(pos, dtype=np.float32).astype(np.float64)     # the kernel reads float32 vertices
    tri = np.asarray(tri, dtype=np.int64)
    v = pos[tri]                                                    # [F,3,4]
    xyw = v[..., [0, 1, 3]]
    x, y, w = xyw[..., 0], xyw[..., 1], xyw[..., 2]
    det = (x[:, 0] * (y[:, 1] * w[:, 2] - y[:, 2] * w[:, 1]) + y[:, 0] * (x[:, 2] * w[:, 1] - x[:, 1] * w[:, 2]) +
           w[:, 0] * (x[:, 1] * y[:, 2] - x[:, 2] * y[:, 1]))
    ok = np.isfinite(det) & (det != 0)
    sgn = np.where(det > 0, 1.0, -1.0)
    E = [_edge_coeffs(xyw[:, 1], xyw[:, 2], sgn), _edge_coeffs(xyw[:, 2], xyw[:, 0], sgn),
         _edge_coeffs(xyw[:, 0], xyw[:, 1], sgn)]
    return v, ok, E


def _owns(e, a, b):
    return (e > 0) | ((e == 0) & ((a > 0) | ((a == 0) & (b > 0))))


def _shade(E, v, f, X, Y):
    """Coverage + attributes of triangles f (array [n]) at sample points X, Y ([n, ...] broadcastable)."""
    ex = []
    inside = True
    bshape = (-1,) + (1,) * (X.ndim - 1)
    for a, b, c in E:
        a, b, c = a[f].reshape(bshape), b[f].reshape(bshape), c[f].reshape(bshape)
        e = a * X + (b * Y + c)
        inside = inside & _owns(e, a, b)
        ex.append(e)
    s = ex[0] + ex[1] + ex[2]
    inside = inside & (s > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = 1.0 / s
        b0, b1, b2 = ex[0] * r, ex[1] * r, ex[2] * r
        z = v[f][:, :, 2]
        w = v[f][:, :, 3]
        zc = b0 * z[:, 0].reshape(bshape) + b1 * z[:, 1].reshape(bshape) + b2 * z[:, 2].reshape(bshape)
        wc = b0 * w[:, 0].reshape(bshape) + b1 * w[:, 1].reshape(bshape) + b2 * w[:, 2].reshape(bshape)
        zw = (zc / wc).astype(np.float32)
    inside = inside & (zw >= -1) & (zw <= 1)
    return inside, b0, b1, zw


def _ordered(zw32):
    u = zw32.view(np.uint32).astype(np.uint64)
    return np.where(u & 0x80000000, (~u) & 0xFFFFFFFF, u | 0x80000000)


def _resolve(H, W, pix, key, b0, b1, zw):
    """Nearest fragment per pixel (ties: smaller id), then its attributes."""
    best = np.full(H * W, np.iinfo(np.uint64).max, dtype=np.uint64)
    np.minimum.at(best, pix, key)
    win = best[pix] == key
    rast = np.zeros((H * W, 4))
    p = pix[win]
    rast[p, 0], rast[p, 1], rast[p, 2] = b0[win], b1[win], zw[win]
    rast[p, 3] = (key[win] & 0xFFFFFFFF).astype(np.float64) + 1
    return rast.reshape(H, W, 4)


def rasterize(pos, tri, height, width):
    """Brute force: every triangle against every pixel.  pos [V,4] clip space, tri [F,3] ->
    rast [H,W,4] float64 (u, v, z/w, id+1)."""
    v, ok, E = _setup(pos, tri)
    H, W = height, width
    X = ((np.arange(W) + 0.5) / W * 2 - 1)[None, None, :]
    Y = ((np.arange(H) + 0.5) / H * 2 - 1)[None, :, None]
    f = np.nonzero(ok)[0]
    if f.size == 0:
        return np.zeros((H, W, 4))
    inside, b0, b1, zw = _shade(E, v, f, X, Y)                     # [n,H,W]
    n, yy, xx = np.nonzero(inside)
    pix = yy * W + xx
    key = (_ordered(zw[n, yy, xx]) << np.uint64(32)) | f[n].astype(np.uint64)
    return _resolve(H, W, pix, key, b0[n, yy, xx], b1[n, yy, xx], zw[n, yy, xx].astype(np.float64))


def rasterize_windowed(pos, tri, height, width, window=12, chunk=20000):
    """Same result as rasterize(); each triangle is evaluated over its pixel bounding box only
    (boxes up to window x window vectorised across triangles, larger ones one by one)."""
    v, ok, E = _setup(pos, tri)
    H, W = height, width
    w = v[:, :, 3]
    front = (w > 1e-6).all(1)
    behind = (w <= 1e-6).all(1)
    with np.errstate(divide="ignore", invalid="ignore"):
        nx, ny = v[:, :, 0] / w, v[:, :, 1] / w
    x0 = np.floor((nx.min(1) + 1) * 0.5 * W - 0.5) - 1
    x1 = np.ceil((nx.max(1) + 1) * 0.5 * W - 0.5) + 2
    y0 = np.floor((ny.min(1) + 1) * 0.5 * H - 0.5) - 1
    y1 = np.ceil((ny.max(1) + 1) * 0.5 * H - 0.5) + 2
    whole = ~front & ~behind                                        # crosses the eye plane: whole screen
    x0 = np.where(whole, 0, x0); y0 = np.where(whole, 0, y0); x1 = np.where(whole, W, x1); y1 = np.where(whole, H, y1)
    off = front & ((nx.max(1) < -1) | (nx.min(1) > 1) | (ny.max(1) < -1) | (ny.min(1) > 1))
    with np.errstate(invalid="ignore"):
        x0 = np.clip(np.nan_to_num(x0), 0, W).astype(np.int64); x1 = np.clip(np.nan_to_num(x1), 0, W).astype(np.int64)
        y0 = np.clip(np.nan_to_num(y0), 0, H).astype(np.int64); y1 = np.clip(np.nan_to_num(y1), 0, H).astype(np.int64)
    live = ok & ~behind & ~off & (x1 > x0) & (y1 > y0)
    small = live & (x1 - x0 <= window) & (y1 - y0 <= window)
    pix_l, key_l, b0_l, b1_l, zw_l = [], [], [], [], []

    def emit(f, inside, b0, b1, zw, px, py):
        sel = np.nonzero(inside)
        fi = f[sel[0]]
        pix_l.append((py[sel] * W + px[sel]).astype(np.int64))
        key_l.append((_ordered(zw[sel]) << np.uint64(32)) | fi.astype(np.uint64))
        b0_l.append(b0[sel]); b1_l.append(b1[sel]); zw_l.append(zw[sel].astype(np.float64))

    fs = np.nonzero(small)[0]
    k = np.arange(window)
    for s in range(0, fs.size, chunk):
        f = fs[s:s + chunk]
        px = x0[f][:, None, None] + k[None, None, :] + 0 * k[None, :, None]
        py = y0[f][:, None, None] + k[None, :, None] + 0 * k[None, None, :]
        valid = (px < x1[f][:, None, None]) & (py < y1[f][:, None, None])
        X = (px + 0.5) / W * 2 - 1
        Y = (py + 0.5) / H * 2 - 1
        inside, b0, b1, zw = _shade(E, v, f, X, Y)
        emit(f, inside & valid, b0, b1, zw, px, py)
    for fi in np.nonzero(live & ~small)[0]:
        f = np.array([fi])
        px, py = np.meshgrid(np.arange(x0[fi], x1[fi]), np.arange(y0[fi], y1[fi]))
        px, py = px[None], py[None]
        inside, b0, b1, zw = _shade(E, v, f, (px + 0.5) / W * 2 - 1, (py + 0.5) / H * 2 - 1)
        emit(f, inside, b0, b1, zw, px, py)
    if not pix_l:
        return np.zeros((H, W, 4))
    return _resolve(H, W, np.concatenate(pix_l), np.concatenate(key_l), np.concatenate(b0_l), np.concatenate(b1_l),
                    np.concatenate(zw_l))


def projected_area_px(pos, tri, height, width):
    """|signed area| of every triangle in pixels^2 (inf when a vertex is not in front of the eye)."""
    pos = np.asarray(pos, dtype=np.float64)
    v = pos[np.asarray(tri, dtype=np.int64)]
    w = v[:, :, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        x = (v[:, :, 0] / w + 1) * 0.5 * width
        y = (v[:, :, 1] / w + 1) * 0.5 * height
    a = 0.5 * np.abs((x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0]))
    return np.where((w > 1e-6).all(1), a, np.inf)

"""CPU checks of the triangle-raster oracle (oracle/mesh_oracle.py): the bounding-box form used at the C4
mesh size against the brute-force form, and the pinned fill rule (top-left on exactly shared edge
functions: a closed surface is covered once, without cracks or double hits)."""
import numpy as np
import torch

from frosting_amd import scenes
from oracle import mesh_oracle as MO


def _clip(verts, cam):
    v = torch.cat([verts, torch.ones(verts.shape[0], 1)], 1) @ cam.projmatrix
    return v.numpy()


def test_windowed_equals_brute_force():
    cam = scenes.ring_camera(1, 96, 64, 80.0, 80.0)
    verts, faces = scenes.sphere_mesh(10, 16)
    g = torch.Generator().manual_seed(3)
    verts = verts + 0.01 * torch.randn(verts.shape, generator=g)
    pos = _clip(verts, cam)
    a = MO.rasterize(pos, faces.numpy(), 64, 96)
    b = MO.rasterize_windowed(pos, faces.numpy(), 64, 96, window=6)     # small window: exercises both paths
    np.testing.assert_array_equal(a, b)
    assert (a[..., 3] > 0).any() and (a[..., 3] == 0).any()


def test_fill_rule_shared_edges_are_covered_exactly_once():
    """A fan of triangles around the image centre whose shared edges pass exactly through pixel centres
    (diagonals, horizontals, verticals of a 16x16 image in NDC multiples of 1/16): every covered pixel is
    claimed by exactly one triangle of the closed fan, with or without the depth test."""
    H = W = 16
    c = 1.0 / 16                                                         # pixel centre (8, 8) sits at NDC (c, c)
    ring = [(-0.5 + c, -0.5 + c), (0.5 + c, -0.5 + c), (0.5 + c, 0.5 + c), (-0.5 + c, 0.5 + c)]
    pos = np.array([[c, c, 0.0, 1.0]] + [[x, y, 0.0, 1.0] for x, y in ring], dtype=np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 4, 1]], dtype=np.int32)
    v, ok, E = MO._setup(pos, tri)
    X = ((np.arange(W) + 0.5) / W * 2 - 1)[None, None, :]
    Y = ((np.arange(H) + 0.5) / H * 2 - 1)[None, :, None]
    inside, *_ = MO._shade(E, v, np.arange(4), X, Y)
    cover = inside.sum(0)
    assert cover.max() == 1                                              # no double hits on the shared edges
    # the open square's interior is fully covered (no cracks along the diagonals / at the hub)
    assert cover[5:12, 5:12].min() == 1 and int(cover.sum()) >= 49
    # the four outer edges lie exactly on pixel centres too: exactly two of them (left, and the b > 0 one) own theirs
    assert int(cover.sum()) == 8 * 8                                      # half-open square: 8 x 8 pixel centres
    # reversing a triangle's winding or relabelling its vertices does not move its coverage
    tri2 = tri[:, [1, 2, 0]].copy(); tri2[1] = tri2[1][[1, 0, 2]]
    v2, ok2, E2 = MO._setup(pos, tri2)
    inside2, *_ = MO._shade(E2, v2, np.arange(4), X, Y)
    np.testing.assert_array_equal(inside, inside2)


def test_contract_of_the_output_planes():
    pos = np.array([[-1.0, -1.0, 0.0, 1.0], [1.0, -1.0, 0.0, 1.0], [0.0, 1.0, 0.5, 1.0],          # big, far-ish
                    [-0.5, -0.5, -0.5, 1.0], [0.5, -0.5, -0.5, 1.0], [0.0, 0.5, -0.5, 1.0],        # nearer
                    [-0.2, 0.0, 2.0, 1.0], [0.2, 0.0, 2.0, 1.0], [0.0, 0.3, 2.0, 1.0]], np.float32)  # beyond the far plane
    tri = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]], np.int32)
    r = MO.rasterize(pos, tri, 32, 32)
    ids = r[..., 3].astype(int)
    assert set(np.unique(ids)) == {0, 1, 2}                              # the clipped triangle never shows
    assert (ids[12:18, 14:18] == 2).all()                                # nearest wins
    cov = ids > 0
    assert (r[cov][:, 0] >= 0).all() and (r[cov][:, 1] >= 0).all() and (r[cov][:, 0] + r[cov][:, 1] <= 1 + 1e-12).all()
    assert not r[~cov].any()
    # row index grows with NDC y: the apex (y = +1) is in the LAST rows
    assert ids[-2:, :].any() and ids[:, :].any() and cov[1].sum() > cov[-2].sum()
    area = MO.projected_area_px(pos, tri, 32, 32)
    assert abs(area[0] - 0.5 * 32 * 32) < 1e-6


def test_depth_keep_mask_semantics():
    """frosting_amd.mesh.depth_keep_mask, the 'depth' culling variant (frosting_model.py:1547-1562): in front of the map (plus a
    tolerance), or no depth there, and inside the image; the map sampled bilinearly at the projected centre with zeros outside
    (a literal four-tap restatement for interior points)."""
    import math
    import torch
    from frosting_amd import mesh as M, scenes
    cam = scenes.ring_camera(0, 64, 48, 60.0, 60.0)
    H, W = 48, 64
    depth = torch.zeros(H, W)
    depth[:, : W // 2] = 5.0                                   # a wall at view depth 5 over the left half, nothing on the right
    depth[10:20, 5:15] = 3.0
    g = torch.Generator().manual_seed(5)
    # points in VIEW space, then to world with the inverse of the (row-vector) view matrix
    zs = torch.rand(400, generator=g) * 8 + 1
    xs = (torch.rand(400, generator=g) * 2.3 - 1.15) * zs * (W / 2) / 60.0
    ys = (torch.rand(400, generator=g) * 2.3 - 1.15) * zs * (H / 2) / 60.0
    pv = torch.stack([xs, ys, zs, torch.ones(400)], 1)
    world = (pv @ torch.linalg.inv(cam.viewmatrix))[:, :3].contiguous()
    tol = 0.25
    keep = M.depth_keep_mask(world, cam.viewmatrix, cam.projmatrix, depth, tol)
    hom = torch.cat([world, torch.ones(400, 1)], 1) @ cam.projmatrix
    ndc = hom[:, :2] / hom[:, 3:4]
    n_in = n_out = n_cut = 0
    for i in range(400):
        x, y = float(ndc[i, 0]), float(ndc[i, 1])
        if abs(x) > 1 or abs(y) > 1:
            assert not bool(keep[i]); n_out += 1
            continue
        px, py = ((x + 1) * W - 1) / 2, ((y + 1) * H - 1) / 2       # ndc2Pix: the pixel-centre convention of align_corners=False
        x0, y0 = math.floor(px), math.floor(py)
        tap = lambda yy, xx: float(depth[yy, xx]) if 0 <= yy < H and 0 <= xx < W else 0.0
        fx, fy = px - x0, py - y0
        mz = (tap(y0, x0) * (1 - fx) + tap(y0, x0 + 1) * fx) * (1 - fy) + (tap(y0 + 1, x0) * (1 - fx) + tap(y0 + 1, x0 + 1) * fx) * fy
        want = (float(zs[i]) < mz + tol) or mz <= 0.0
        if abs(float(zs[i]) - (mz + tol)) > 1e-3 and abs(mz) > 1e-6:   # away from the decision boundaries (float32 sampling)
            assert bool(keep[i]) == want, (i, float(zs[i]), mz)
            n_in += 1
            n_cut += int(not want)
    assert n_out > 20 and n_in > 150 and n_cut > 30

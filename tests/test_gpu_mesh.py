"""GPU tests of the triangle occlusion raster against the CPU
z-buffer restatement (oracle/mesh_oracle.py; parity for this third-party side op is
unpinned -- see that file's header)."""
import math

import numpy as np
import pytest
import torch

from frosting_amd import mesh as M
from frosting_amd import scenes
from oracle import mesh_oracle as MO

pytestmark = pytest.mark.gpu


def sphere_mesh(n_lat, n_lon, radius=1.0):
    return scenes.sphere_mesh(n_lat, n_lon, radius)


def _compare_ids(got_ids, ref_ids, pos, faces, H, W, label):
    """SURVEY 8(c) acceptance: identical visible-face sets up to faces covering less than a pixel.
    Returns (pixels that differ, faces in exactly one set)."""
    diff_px = int((got_ids != ref_ids).sum())
    a, b = set(np.unique(got_ids).tolist()), set(np.unique(ref_ids).tolist())
    only = sorted((a ^ b) - {0})
    area = MO.projected_area_px(pos, faces, H, W)
    print(f"\n[{label}] pixels with a different face id: {diff_px} of {H * W}; faces visible in only one of the two: "
          f"{len(only)} (projected areas {[round(float(area[f - 1]), 3) for f in only[:8]]} px^2)")
    assert all(area[f - 1] < 1.0 for f in only), "a face covering a pixel or more is visible in only one rasterizer"
    return diff_px, len(only)


def test_matches_cpu_zbuffer_small_mesh(gpu_device):
    cam = scenes.ring_camera(1, 96, 64, 80.0, 80.0)
    verts, faces = sphere_mesh(10, 16)
    g = torch.Generator().manual_seed(3)
    verts = verts + 0.01 * torch.randn(verts.shape, generator=g)
    pos = M.clip_space_vertices(verts.to(gpu_device), cam.projmatrix.to(gpu_device))
    rast, _ = M.rasterize(M.RasterizeGLContext(), pos, faces.to(gpu_device), [64, 96])
    ref = MO.rasterize(pos[0].cpu().numpy(), faces.numpy(), 64, 96)
    got = rast[0].cpu().numpy()
    ids, rids = got[..., 3].astype(np.int64), ref[..., 3].astype(np.int64)
    # same float64 edge functions, same fill rule, same depth rule: the id planes are identical
    diff_px, only = _compare_ids(ids, rids, pos[0].cpu().numpy(), faces.numpy(), 64, 96, "sphere 320 tris, 96x64")
    assert diff_px == 0 and only == 0
    np.testing.assert_allclose(got[..., :3], ref[..., :3], atol=2e-6)
    covered = ids > 0
    assert covered.any() and (~covered).any()
    assert not got[~covered].any()                    # empty pixels are all-zero
    b = got[covered]
    assert (b[:, 0] >= 0).all() and (b[:, 1] >= 0).all() and (b[:, 0] + b[:, 1] <= 1 + 1e-6).all()


def test_fill_rule_on_exactly_shared_edges(gpu_device):
    """Edges through pixel centres (all coordinates exactly representable): top-left ownership, every centre of
    the closed fan covered exactly once -- the id plane equals the oracle's bit for bit, for both windings."""
    H = W = 16
    c = 1.0 / 16
    ring = [(-0.5 + c, -0.5 + c), (0.5 + c, -0.5 + c), (0.5 + c, 0.5 + c), (-0.5 + c, 0.5 + c)]
    pos = torch.tensor([[c, c, 0.0, 1.0]] + [[x, y, 0.0, 1.0] for x, y in ring])
    for tri in ([[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 4, 1]], [[1, 0, 2], [3, 2, 0], [0, 3, 4], [4, 1, 0]]):
        tri = torch.tensor(tri, dtype=torch.int32)
        rast, _ = M.rasterize(None, pos[None].to(gpu_device), tri.to(gpu_device), [H, W])
        ids = rast[0, ..., 3].cpu().numpy().astype(int)
        ref = MO.rasterize(pos.numpy(), tri.numpy(), H, W)[..., 3].astype(int)
        np.testing.assert_array_equal(ids, ref)
        assert int((ids > 0).sum()) == 64 and set(np.unique(ids)) == {0, 1, 2, 3, 4}


def test_c4_mesh_visible_faces_match_the_oracle_at_full_size(gpu_device):
    """BASELINE configs[3]'s occlusion mesh at full size: 200 704 triangles, 1600x1056, camera 0.
    unique(pix_to_face) must equal the float64 oracle's except for faces whose projection is below one
    pixel; the number of differing pixels and faces is reported."""
    cfg = scenes.CONFIGS["c4"]
    verts, faces = scenes.sphere_mesh(cfg["n_lat"], cfg["n_lon"])
    assert faces.shape[0] == 200_704
    H, W = cfg["height"], cfg["width"]
    for view in (0, 3):
        cam = scenes.ring_camera(view, W, H, cfg["fx"], cfg["fy"])
        pos = M.clip_space_vertices(verts.to(gpu_device), cam.projmatrix.to(gpu_device))
        rast, _ = M.rasterize(M.RasterizeGLContext(), pos, faces.to(gpu_device), [H, W])
        ids = rast[0, ..., 3].cpu().numpy().astype(np.int64)
        ref = MO.rasterize_windowed(pos[0].cpu().numpy(), faces.numpy(), H, W)
        rids = ref[..., 3].astype(np.int64)
        diff_px, only = _compare_ids(ids, rids, pos[0].cpu().numpy(), faces.numpy(), H, W, f"C4 mesh, view {view}")
        assert diff_px <= 20                      # depth ties on the silhouette at most; measured 0
        same = ids == rids
        np.testing.assert_allclose(rast[0].cpu().numpy()[..., :3][same], ref[..., :3][same], atol=5e-6)
        vis = M.visible_faces(verts.to(gpu_device), faces.to(gpu_device), cam.projmatrix.to(gpu_device), H, W)
        frac = vis.numel() / faces.shape[0]
        assert 0.2 < frac < 0.45                  # by COUNT (lat-long faces crowd the poles); ~37 % by area
        # the culling path's form (frg_mesh_visible_faces: z-buffer + one byte per covered pixel, no id plane): the same set
        fm = M.visible_face_mask(verts.to(gpu_device), faces.to(gpu_device), cam.projmatrix.to(gpu_device), H, W)
        assert torch.equal(torch.nonzero(fm).flatten(), vis.sort().values)
        assert set(np.unique(ids[ids > 0]) - 1) == set(torch.nonzero(fm).flatten().cpu().numpy())


def test_fragments_contract_on_the_c4_mesh(gpu_device):
    """The second consumer of the triangle raster in the reference: MeshRasterizer.forward's nvdiffrast branch turns
    dr.rasterize's plane into pytorch3d `Fragments` (frosting_utils/mesh_rasterization.py:146-169) -- pix_to_face (-1 =
    empty), bary_coords with the third coordinate 1 - u - v, zbuf = z/w.  frosting_amd.mesh.fragments on the C4 mesh
    (200 704 triangles, 1600x1056) against what those planes MEAN: in every covered pixel the barycentric combination
    of the face's clip-space vertices lands on the pixel centre (perspective-correct weights: u, v weigh the
    homogeneous vertices) and reproduces zbuf, the weights are a partition of unity inside the triangle, and the face is
    the nearest one over the float64 oracle's plane on a 64 x 48 window (round 3 checked the shim on one triangle)."""
    shell, cam, _ = scenes.config_shell_scene("c4", 3, P=1000)
    dev = gpu_device
    H, W = cam.image_height, cam.image_width
    pos = M.clip_space_vertices(shell.verts.to(dev), cam.projmatrix.to(dev))
    faces = shell.faces.to(dev)
    bary, zbuf, p2f = M.fragments(M.RasterizeGLContext(), pos, faces, [H, W])
    assert bary.shape == (1, H, W, 1, 3) and zbuf.shape == (1, H, W, 1) and p2f.shape == (1, H, W, 1) and p2f.dtype == torch.int32
    cov = p2f[0, :, :, 0] >= 0
    assert 0.2 < float(cov.float().mean()) < 0.6 and int(p2f.min()) == -1 and int(p2f.max()) < faces.shape[0]
    ys, xs = torch.nonzero(cov, as_tuple=True)
    f = p2f[0, ys, xs, 0].long()
    b = bary[0, ys, xs, 0].double()                                     # [n, 3]
    assert float(b.min()) > -1e-5 and float((b.sum(1) - 1).abs().max()) < 1e-6
    v = pos[0].double()[faces[f].long()]                                # [n, 3, 4] clip-space vertices of the hit face
    # nvdiffrast's (u, v) are perspective-correct weights of vertices 0 and 1: the homogeneous point is their combination
    hom = (b[:, :, None] * v).sum(1)
    ndc = hom[:, :3] / hom[:, 3:4]
    px = (ndc[:, 0] + 1.0) * 0.5 * W - 0.5
    py = (ndc[:, 1] + 1.0) * 0.5 * H - 0.5
    assert float((px - xs.double()).abs().max()) < 2e-3 and float((py - ys.double()).abs().max()) < 2e-3
    assert float((ndc[:, 2] - zbuf[0, ys, xs, 0].double()).abs().max()) < 2e-6
    # nearest face: the float64 z-buffer oracle on a window of the image
    y0, x0 = H // 2 - 24, W // 2 - 32
    want = MO.rasterize_windowed(pos[0].cpu().numpy(), shell.faces.numpy(), H, W)[y0:y0 + 48, x0:x0 + 64, 3].astype(np.int64) - 1
    assert np.array_equal(p2f[0, y0:y0 + 48, x0:x0 + 64, 0].cpu().numpy().astype(np.int64), want)


def test_depth_order_near_plane_and_big_triangles(gpu_device):
    dev = gpu_device
    # two screen-filling triangles at different depths plus one crossing w = 0
    pos = torch.tensor([[-3.0, -3.0, 0.5, 1.0], [3.0, -3.0, 0.5, 1.0], [0.0, 3.0, 0.5, 1.0],      # far, covers all
                        [-0.5, -0.5, 0.2, 1.0], [0.5, -0.5, 0.2, 1.0], [0.0, 0.5, 0.2, 1.0],      # near, small
                        [-0.2, 0.0, 0.1, 0.5], [0.2, 0.0, 0.1, 0.5], [0.0, 0.3, -0.2, -0.4]],     # crosses the eye plane
                       device=dev)
    tri = torch.tensor([[0, 1, 2], [3, 4, 5], [6, 7, 8]], dtype=torch.int32, device=dev)
    rast, _ = M.rasterize(None, pos[None], tri, [48, 64])
    ref = MO.rasterize(pos.cpu().numpy(), tri.cpu().numpy(), 48, 64)
    ids, rids = rast[0, ..., 3].cpu().numpy().astype(int), ref[..., 3].astype(int)
    assert (ids == rids).mean() > 0.999
    assert (ids > 0).all()                      # the far triangle covers the whole image
    assert (ids == 2).sum() > 50                # the near one wins where it is
    vis = M.visible_faces(pos[:, :3] * 0 + pos[:, :3], tri, torch.eye(4, device=dev), 48, 64)
    assert set(vis.tolist()) <= {0, 1, 2}


def test_occlusion_culling_mask_c4_shape(gpu_device):
    """BASELINE config 4 plumbing at reduced size: shell Gaussians bound to the faces of a
    sphere, culling = faces visible from the camera (about the front hemisphere)."""
    cam = scenes.ring_camera(0, 400, 264, 333.5, 333.5)
    verts, faces = sphere_mesh(60, 120)          # 14400 triangles
    dev = gpu_device
    vis = M.visible_faces(verts.to(dev), faces.to(dev), cam.projmatrix.to(dev), cam.image_height, cam.image_width)
    F = faces.shape[0]
    frac = vis.numel() / F
    assert 0.25 < frac < 0.55                    # ~37 % of the sphere area is visible from distance 4
    # every visible face must face the camera
    v = verts[faces.long()]
    n = torch.linalg.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    centre = v.mean(1)
    facing = ((cam.campos[None] - centre) * centre).sum(1) > -1e-3   # outward = centre direction on a sphere
    assert facing[vis.cpu()].float().mean() > 0.99
    P = 20000
    g = torch.Generator().manual_seed(0)
    cell = torch.randint(0, F, (P,), generator=g).to(dev)
    keep = M.occlusion_mask(cell, vis, F, n_background=100)
    assert keep.shape[0] == P + 100 and keep[-100:].all()
    # the boolean-mask route (no torch.unique) selects the same faces and Gaussians
    fm = M.visible_face_mask(verts.to(dev), faces.to(dev), cam.projmatrix.to(dev), cam.image_height, cam.image_width)
    assert torch.equal(torch.nonzero(fm).flatten(), vis.sort().values)
    assert torch.equal(M.occlusion_mask_from_face_mask(cell, fm, n_background=100), keep)
    assert abs(keep[:P].float().mean().item() - frac) < 0.02


def test_occlusion_keep_mask_in_one_call_equals_the_composition(gpu_device):
    """frg_mesh_occlusion_mask (vertex transform + clears, z-buffer, marks, gather in five launches) against the composition it
    replaces -- torch.ones / cat / matmul, frg_mesh_visible_faces, the torch index and the background ones
    (frosting_model.py:1524-1539, 1564-1586): the same faces and the same keep mask on the full C4 mesh from two cameras,
    on a small mesh with background Gaussians and negative cell indices, and on the degenerate sizes.  The clip-space
    vertices it forms differ from the matmul's in no bit (reported if they ever do: the mask is what is asserted)."""
    dev = gpu_device
    cfg = scenes.CONFIGS["c4"]
    verts, faces = scenes.sphere_mesh(cfg["n_lat"], cfg["n_lon"])
    H, W = cfg["height"], cfg["width"]
    g = torch.Generator().manual_seed(3)
    cell = torch.randint(0, faces.shape[0], (500_000,), generator=g)
    ctx = M.RasterizeGLContext()
    for view in (0, 5):
        cam = scenes.ring_camera(view, W, H, cfg["fx"], cfg["fy"])
        fm = M.visible_face_mask(verts.to(dev), faces.to(dev), cam.projmatrix.to(dev), H, W)
        want = M.occlusion_mask_from_face_mask(cell.to(dev), fm, n_background=77)
        keep, fm1 = M.occlusion_keep_mask(verts.to(dev), faces.to(dev), cam.projmatrix.to(dev), H, W, cell.to(dev), 77, ctx, return_face_mask=True)
        assert keep.dtype == torch.bool and keep.shape == want.shape
        ndiff = int((fm1 != fm).sum())
        assert ndiff == 0, f"view {view}: {ndiff} faces differ between the fused and the composed path"
        assert torch.equal(keep, want) and keep[-77:].all()
        assert 0.2 < float(keep[:-77].float().mean()) < 0.45
        # the clip-space vertices it formed (behind the raster's part of the workspace) against the matmul's
        from frosting_amd import _lib
        V, F = verts.shape[0], faces.shape[0]
        off = int(_lib.lib().frg_mesh_raster_workspace_bytes(F, W, H))
        p4 = ctx._work[off:off + V * 16].view(torch.float32).view(V, 4)
        pos = M.clip_space_vertices(verts.to(dev), cam.projmatrix.to(dev))[0]
        nbits = int((p4 != pos).sum())
        print(f"view {view}: clip-space vertices, fused multiply-adds vs matmul: {nbits} of {4 * V} floats differ"
              + (f", max |diff| {float((p4 - pos).abs().max()):.3g}" if nbits else ""))
        assert torch.allclose(p4, pos, rtol=2e-6, atol=1e-6)
    # a small mesh, cells given from the end (torch's negative indices), no background
    cam = scenes.ring_camera(2, 400, 264, 333.5, 333.5)
    v, f = sphere_mesh(40, 80)
    cells = torch.randint(-f.shape[0], f.shape[0], (10_000,), generator=g)
    fm = M.visible_face_mask(v.to(dev), f.to(dev), cam.projmatrix.to(dev), 264, 400)
    keep = M.occlusion_keep_mask(v.to(dev), f.to(dev), cam.projmatrix.to(dev), 264, 400, cells.to(dev), 0)
    assert torch.equal(keep, fm[cells.to(dev)])
    # no Gaussians; no faces
    assert M.occlusion_keep_mask(v.to(dev), f.to(dev), cam.projmatrix.to(dev), 264, 400, cells[:0].to(dev), 0).numel() == 0
    none = M.occlusion_keep_mask(v.to(dev), f[:0].to(dev), cam.projmatrix.to(dev), 264, 400, cells[:0].to(dev), 5)
    assert none.shape[0] == 5 and none.all()


def test_nvdiffrast_module_shim(gpu_device):
    M.install_as_nvdiffrast()
    import nvdiffrast.torch as dr
    ctx = dr.RasterizeGLContext()
    pos = torch.tensor([[[-1.0, -1.0, 0.0, 1.0], [1.0, -1.0, 0.0, 1.0], [0.0, 1.0, 0.0, 1.0]]], device=gpu_device)
    rast_out, _ = dr.rasterize(ctx, pos=pos, tri=torch.tensor([[0, 1, 2]], dtype=torch.int32, device=gpu_device),
                               resolution=[32, 32])
    bary_coords, zbuf, pix_to_face = rast_out[..., :2], rast_out[..., 2], rast_out[..., 3].int()   # nvdiffrast.py:54
    assert rast_out.shape == (1, 32, 32, 4) and (pix_to_face.unique() - 1).tolist() == [-1, 0]


def test_keep_mask_equals_boolean_compaction(gpu_device):
    """Occlusion culling as a skip flag (frg_forward_ex keep_mask) against the reference's way --
    boolean compaction of every per-Gaussian tensor before the render
    (frosting_scene/frosting_model.py:1564-1586): same image and radii bit for bit, same gradients."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from helpers import settings_for
    dev = gpu_device
    scene, cam, bg = scenes.config_scene("mini", 4, P=6000)
    g = torch.Generator().manual_seed(11)
    keep = (torch.rand(scene.P, generator=g) < 0.6).to(dev)
    sc = scene.to(dev)
    rast = GaussianRasterizer(settings_for(cam, bg, scene.sh_degree, dev))
    names = ("means3D", "opacities", "shs", "scales", "rotations")

    def leaves(sel):
        out = {}
        for n in names:
            t = getattr(sc, n)
            out[n] = (t[sel] if sel is not None else t).clone().requires_grad_(True)
        return out

    a = leaves(keep)                       # the reference's way: compact, then render
    m2d_a = torch.zeros_like(a["means3D"], requires_grad=True)
    img_a, radii_a = rast(means3D=a["means3D"], means2D=m2d_a, opacities=a["opacities"], shs=a["shs"],
                          scales=a["scales"], rotations=a["rotations"])
    b = leaves(None)                       # ours: full tensors + skip flag
    m2d_b = torch.zeros_like(b["means3D"], requires_grad=True)
    img_b, radii_b = rast(means3D=b["means3D"], means2D=m2d_b, opacities=b["opacities"], shs=b["shs"],
                          scales=b["scales"], rotations=b["rotations"], keep_mask=keep)
    assert torch.equal(img_a, img_b)
    assert torch.equal(radii_b[keep], radii_a) and int(radii_b[~keep].abs().max()) == 0
    assert int((radii_a > 0).sum()) > 100
    gpix, _ = scenes.l1_target_grad(img_a.detach().cpu(), 5)
    gpix = gpix.to(dev)
    img_a.backward(gpix)
    img_b.backward(gpix)
    # gradients: same sums, but the per-Gaussian reduction groups 64 consecutive Gaussians per wave, so
    # its association order follows the indexing (compacted vs full) -- equal to float32 rounding
    from helpers import rel_l2
    for n in names:
        assert rel_l2(b[n].grad[keep], a[n].grad) < 5e-5, n
        assert float(b[n].grad[~keep].abs().max()) == 0.0, n
    assert rel_l2(m2d_b.grad[keep], m2d_a.grad) < 5e-5
    # uint8 masks work too; a bad shape is rejected
    img_c, _ = rast(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), opacities=sc.opacities, shs=sc.shs,
                    scales=sc.scales, rotations=sc.rotations, keep_mask=keep.to(torch.uint8))
    assert torch.equal(img_c, img_a)
    with pytest.raises(RuntimeError, match="keep_mask"):
        rast(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), opacities=sc.opacities, shs=sc.shs,
             scales=sc.scales, rotations=sc.rotations, keep_mask=keep[:-1])


@pytest.mark.skipif(not __import__("oracle.ref_rasterizer", fromlist=["x"]).available("exact"), reason="oracle/_ref not built")
@pytest.mark.parametrize("P", [2_000_000])
def test_c4_refine_step_vs_reference_on_compacted_tensors(gpu_device, P):
    """BASELINE configs[3] end to end at full size: HIP mesh raster -> visible-face mask -> keep_mask render
    (skip flag inside preprocess), against the REFERENCE rasterizer fed the boolean-compacted tensors the way
    frosting_model.py:1564-1586 builds them.  EXACT mode: image and radii bit-identical; gradients at the
    reference's own noise."""
    import helpers as Hh
    from frosting_amd import _lib
    from oracle import ref_rasterizer as REF
    import diff_gaussian_rasterization as D
    dev = gpu_device
    shell, cam, bg = scenes.config_shell_scene("c4", 0, P=P)
    sh = shell.to(dev)
    sc = sh.scene
    H, W = cam.image_height, cam.image_width
    fm = M.visible_face_mask(sh.verts, sh.faces, cam.projmatrix.to(dev), H, W)
    keep = M.occlusion_mask_from_face_mask(sh.cell, fm)
    assert 0.25 < float(keep.float().mean()) < 0.5
    e = torch.Tensor([])
    args = (bg.to(dev), sc.means3D, e, sc.opacities, sc.scales, sc.rotations, 1.0, e, cam.viewmatrix.to(dev),
            cam.projmatrix.to(dev), cam.tanfovx, cam.tanfovy, H, W, sc.shs, 3, cam.campos.to(dev), False, False)
    _lib.set_option("exact_blend", 1)
    try:
        R, color, radii, geom, binning, img = D._C.rasterize_gaussians_masked(*args, keep)
        comp = scenes.Scene(sc.means3D[keep], sc.scales[keep], sc.rotations[keep], sc.opacities[keep], sc.shs[keep], 3)
        Rr, rcolor, rradii, rst = REF.forward(**Hh.oracle_kwargs(comp, cam.to(dev), bg.to(dev), as_numpy=False, device=dev))
        assert R == Rr
        assert torch.equal(radii[keep], rradii) and not radii[~keep].any()
        assert torch.equal(color, rcolor)
        gpix, _ = scenes.l1_target_grad(color.cpu(), 41)
        gpix = gpix.to(dev)
        b = (args[0], args[1], radii, args[2], args[4], args[5], args[6], args[7], args[8], args[9], args[10], args[11],
             gpix, args[14], args[15], args[16], geom, R, binning, img, False)
        grads = D._C.rasterize_gaussians_backward(*b)
        runs = Hh.reference_runs(lambda: REF.backward(rst, gpix))
        for name, g in zip(Hh.GRAD_NAMES, grads):
            assert not g[~keep].any(), name                    # culled Gaussians: zero rows
        # the kept rows beside the reference's four runs on the compacted tensors, against the float64 gradient of that
        # forward state (round 3: floors x 3 and a flat 1e-3 on dL_dmeans3D for this thin shell seen edge-on)
        truth = Hh.truth_from_ref_state(rst, gpix)
        Hh.judge_gradients({name: g[keep] for name, g in zip(Hh.GRAD_NAMES, grads)}, runs, truth, fast=False, label="c4 culled refine step")
        # ... and in the arithmetic bench.py's `c4` line runs: the default blend, with the frame's mask
        # from the one-call occlusion culling (mesh.occlusion_keep_mask -> frg_mesh_occlusion_mask) -- image within the 1e-4
        # per-pixel L1 of north_star, radii and instance count equal, gradients judged as everywhere
        _lib.set_option("exact_blend", 0)
        keep2 = M.occlusion_keep_mask(sh.verts, sh.faces, cam.projmatrix.to(dev), H, W, sh.cell)
        assert torch.equal(keep2.bool(), keep.bool())
        R2, color2, radii2, geom2, binning2, img2 = D._C.rasterize_gaussians_masked(*args, keep2)
        assert R2 == Rr and torch.equal(radii2[keep], rradii) and not radii2[~keep].any()
        assert float((color2 - rcolor).abs().mean()) <= 1e-4
        b2 = b[:2] + (radii2,) + b[3:16] + (geom2, R2, binning2, img2, False)
        grads2 = D._C.rasterize_gaussians_backward(*b2)
        for name, g in zip(Hh.GRAD_NAMES, grads2):
            assert not g[~keep].any(), name
        # (guard=False: the per-tensor regression records of helpers.WELL_OURS_MAX were taken in round 4, before this leg
        # existed -- dL_dcolors sits at 7.5e-6 on this thin shell, 3 x that record and 13 x inside the gate; the gate, the
        # factor on the reference's own distance and the default arithmetic's floor all apply)
        Hh.judge_gradients({name: g[keep] for name, g in zip(Hh.GRAD_NAMES, grads2)}, runs, truth, fast=True,
                           label="c4 culled refine step, default arithmetic", guard=False)
    finally:
        _lib.set_option("exact_blend", 0)

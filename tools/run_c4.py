"""BASELINE config 4 shape: 2M shell-bound Gaussians + 200k-triangle mesh occlusion culling,
1600x1056, one MI355X.  Synthetic stand-in per SURVEY.md 8(d): lat-long unit sphere with
~200k triangles; Gaussians = area-weighted face pick + Dirichlet(1,1,1) barycentrics +
normal offset U(-0.02, 0.02); culling = faces visible from the camera.  Prints timings."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from frosting_amd import mesh as M, scenes, _lib
from frosting_amd.rasterizer import _C

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n_lat, n_lon = 224, 448                      # 2*224*448 = 200 704 triangles
th = torch.linspace(0, math.pi, n_lat + 1, dtype=torch.float64)
ph = torch.linspace(0, 2 * math.pi, n_lon + 1, dtype=torch.float64)[:-1]
T, Pp = torch.meshgrid(th, ph, indexing="ij")
verts = torch.stack([torch.sin(T) * torch.cos(Pp), torch.cos(T), torch.sin(T) * torch.sin(Pp)], -1).reshape(-1, 3).float()
i, j = torch.meshgrid(torch.arange(n_lat), torch.arange(n_lon), indexing="ij")
a, b = (i * n_lon + j).reshape(-1), (i * n_lon + (j + 1) % n_lon).reshape(-1)
c, d = ((i + 1) * n_lon + j).reshape(-1), ((i + 1) * n_lon + (j + 1) % n_lon).reshape(-1)
faces = torch.cat([torch.stack([a, c, b], 1), torch.stack([b, c, d], 1)]).int()
F = faces.shape[0]
g = torch.Generator().manual_seed(scenes.SEED_BASE + 4)
v = verts[faces.long()]
area = torch.linalg.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]).norm(dim=1).double()
cell = torch.multinomial(area / area.sum(), P, replacement=True, generator=g)
bary = torch.distributions.Dirichlet(torch.ones(3)).sample((P,))
pts = (v[cell] * bary[:, :, None]).sum(1)
nrm = torch.nn.functional.normalize(pts, dim=1)
pts = pts + nrm * (torch.rand(P, 1, generator=g) * 0.04 - 0.02)
sc = scenes.make_scene(P, scenes.SEED_BASE + 4)
scene = scenes.Scene(pts.float().contiguous(), sc.scales, sc.rotations, sc.opacities, sc.shs, 3).to(dev)
cam = scenes.ring_camera(0, 1600, 1056, 1334.0, 1334.0).to(dev)
bg = torch.zeros(3, device=dev)
verts_d, faces_d, cell_d = verts.to(dev), faces.to(dev), cell.to(dev)
ctx = M.RasterizeGLContext()

def step(fused):
    e = torch.Tensor([])
    if fused:   # face mask by one index_put (no unique), occlusion mask as a skip flag inside preprocess
        fm = M.visible_face_mask(verts_d, faces_d, cam.projmatrix, 1056, 1600, ctx)
        keep = M.occlusion_mask_from_face_mask(cell_d, fm)
        vis = fm
    else:
        vis = M.visible_faces(verts_d, faces_d, cam.projmatrix, 1056, 1600, ctx)
        keep = M.occlusion_mask(cell_d, vis, F)
    if fused:
        args = (bg, scene.means3D, e, scene.opacities, scene.scales, scene.rotations, 1.0, e,
                cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy, 1056, 1600, scene.shs, 3, cam.campos, False, False)
        out = _C.rasterize_gaussians(*args, keep_mask=keep)
    else:       # the reference's way: boolean compaction of five per-Gaussian tensors, then render
        args = (bg, scene.means3D[keep], e, scene.opacities[keep], scene.scales[keep], scene.rotations[keep], 1.0, e,
                cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy, 1056, 1600, scene.shs[keep], 3, cam.campos, False, False)
        out = _C.rasterize_gaussians(*args)
    return vis, keep, out, args

imgs = {}
for fused in (False, True):
    vis, keep, out, args = step(fused)
    imgs[fused] = out[1].clone()
    gpix = (torch.sign(out[1] - 0.5) / out[1].numel())
    torch.cuda.synchronize()
    tm, tr, tb = [], [], []
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        vis = M.visible_faces(verts_d, faces_d, cam.projmatrix, 1056, 1600, ctx)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        vis, keep, out, args = step(fused)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        R, color, radii, geom, binning, img = out
        bargs = (args[0], args[1], radii, args[2], args[4], args[5], args[6], args[7], args[8], args[9], args[10], args[11],
                 gpix, args[14], args[15], args[16], geom, R, binning, img, False)
        _C.rasterize_gaussians_backward(*bargs)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        tm.append(t1 - t0); tr.append(t2 - t1); tb.append(t3 - t2)
    nvis = int(vis.sum()) if vis.dtype == torch.bool else vis.numel()
    print(f"C4 ({'face mask + skip flag' if fused else 'unique + compaction'}): P={P} tris={F} visible faces {nvis} ({nvis/F:.3f}) "
          f"kept Gaussians {int(keep.sum())} R={out[0]}")
    print(f"  mesh raster + unique: {1e3*np.median(tm):.3f} ms ; raster + cull + forward: {1e3*np.median(tr):.3f} ms ; "
          f"backward: {1e3*np.median(tb):.3f} ms")
print("images identical:", bool(torch.equal(imgs[False], imgs[True])))

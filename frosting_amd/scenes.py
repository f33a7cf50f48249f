"""Deterministic synthetic scenes and cameras for tests and bench.py.

The generator is the one SURVEY.md section 8(d) specifies: a unit-sigma ball of
Gaussians, log-normal scales, SH degree 3, eight ring cameras at distance 4
looking at the origin.  Camera matrices follow the reference's row-vector
convention (frosting_scene/cameras.py:203-212, frosting_utils/graphics_utils.py:52-85):
``world_view_transform`` is the transposed world-to-camera matrix,
``full_proj_transform = world_view @ projection^T`` and the camera centre is row 3
of the inverse world-view matrix.

Everything is produced on the CPU from a seeded ``torch.Generator`` so that the
very same tensors can be fed to the HIP path, the oracle and the reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

SEED_BASE = 20241022


@dataclass
class Camera:
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    viewmatrix: torch.Tensor  # [4,4] world_view_transform (row-vector convention)
    projmatrix: torch.Tensor  # [4,4] full_proj_transform
    campos: torch.Tensor  # [3]

    def to(self, device):
        return Camera(self.image_height, self.image_width, self.tanfovx, self.tanfovy,
                      self.viewmatrix.to(device), self.projmatrix.to(device), self.campos.to(device))


def projection_matrix(znear: float, zfar: float, tanfovx: float, tanfovy: float) -> torch.Tensor:
    """OpenGL-style perspective matrix with z in [0,1] and +z forward
    (frosting_utils/graphics_utils.py:64-85), column-vector form."""
    top, right = tanfovy * znear, tanfovx * znear
    m = torch.zeros(4, 4, dtype=torch.float64)
    m[0, 0] = znear / right
    m[1, 1] = znear / top
    m[3, 2] = 1.0
    m[2, 2] = zfar / (zfar - znear)
    m[2, 3] = -(zfar * znear) / (zfar - znear)
    return m


def look_at_camera(center, width: int, height: int, fx: float, fy: float,
                   znear: float = 0.01, zfar: float = 100.0, principal=None) -> Camera:
    """COLMAP axes (x right, y down, z forward), looking at the origin.
    principal = (cx, cy): an off-centre principal point in NDC units, patched into the projection the way Frosting's
    camera conversion does it (frosting_scene/frosting_model.py:1440-1442: proj[2,0] = -K[0,2], proj[2,1] = -K[1,2])."""
    c = torch.as_tensor(center, dtype=torch.float64)
    z = -c / c.norm()
    x = torch.linalg.cross(z, torch.tensor([0.0, -1.0, 0.0], dtype=torch.float64))
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    w2c = torch.eye(4, dtype=torch.float64)
    w2c[:3, :3] = torch.stack([x, y, z])
    w2c[:3, 3] = -(w2c[:3, :3] @ c)
    tanfovx, tanfovy = width / (2.0 * fx), height / (2.0 * fy)
    # float64 throughout, rounded once (host-independent bits; the reference does the same
    # algebra in float32: full_proj = world_view @ projection, campos = inverse(world_view)[3,:3])
    wv64 = w2c.t().contiguous()  # row-vector convention
    proj64 = projection_matrix(znear, zfar, tanfovx, tanfovy).t().contiguous()
    if principal is not None:
        proj64[2, 0] = -float(principal[0])
        proj64[2, 1] = -float(principal[1])
    world_view = wv64.float()
    full = (wv64 @ proj64).float()
    campos = c.float().contiguous()
    return Camera(height, width, float(tanfovx), float(tanfovy), world_view, full.contiguous(), campos)


def ring_camera(k: int, width: int, height: int, fx: float, fy: float, n: int = 8, dist: float = 4.0, principal=None) -> Camera:
    th = 2.0 * math.pi * k / n
    return look_at_camera([dist * math.sin(th), 0.0, -dist * math.cos(th)], width, height, fx, fy, principal=principal)


@dataclass
class Scene:
    means3D: torch.Tensor  # [P,3]
    scales: torch.Tensor  # [P,3]  (already activated)
    rotations: torch.Tensor  # [P,4]  unit quaternions (r,x,y,z)
    opacities: torch.Tensor  # [P,1]  (already activated)
    shs: torch.Tensor  # [P,16,3]
    sh_degree: int

    def to(self, device):
        return Scene(*(t.to(device) for t in (self.means3D, self.scales, self.rotations, self.opacities, self.shs)),
                     self.sh_degree)

    @property
    def P(self):
        return self.means3D.shape[0]


def make_scene(P: int, seed: int, sh_degree: int = 3, log_scale: float = math.log(0.005),
               scale_sigma: float = 0.8) -> Scene:
    # Everything is drawn and transformed in float64 and rounded once to float32, so the
    # bits do not depend on the host's vector-math library (exp / sqrt / division differ
    # by an ulp between CPU families in float32, which would leak into golden fixtures).
    g = torch.Generator().manual_seed(seed)
    f64 = torch.float64
    means = torch.randn(P, 3, generator=g, dtype=f64)
    nrm = means.norm(dim=1, keepdim=True).clamp_min(1e-12)
    means = torch.where(nrm > 3.0, means * (3.0 / nrm), means).float()
    scales = torch.exp(log_scale + scale_sigma * torch.randn(P, 3, generator=g, dtype=f64)).float()
    q = torch.randn(P, 4, generator=g, dtype=f64)
    q = (q / q.norm(dim=1, keepdim=True)).float()
    opac = (0.05 + 0.9 * torch.rand(P, 1, generator=g, dtype=f64)).float()
    K = 16
    shs = torch.empty(P, K, 3)
    shs[:, 0, :] = (0.5 * torch.randn(P, 3, generator=g, dtype=f64)).float()
    shs[:, 1:, :] = (0.05 * torch.randn(P, K - 1, 3, generator=g, dtype=f64)).float()
    return Scene(means.contiguous(), scales.contiguous(), q.contiguous(), opac.contiguous(), shs.contiguous(), sh_degree)


# BASELINE.json configs (C2 / C3 shapes)
CONFIGS = {
    # small-image case for committed golden fixtures and CPU-sized tests (same FoV as c3)
    "mini": dict(P=3000, width=160, height=112, fx=133.4, fy=133.4, seed=SEED_BASE + 9, bg=(0.1, 0.2, 0.3),
                 log_scale=math.log(0.04)),
    "c2": dict(P=100_000, width=800, height=800, fx=1111.0, fy=1111.0, seed=SEED_BASE + 2, bg=(1.0, 1.0, 1.0)),
    "c3": dict(P=3_000_000, width=1600, height=1056, fx=1334.0, fy=1334.0, seed=SEED_BASE + 3, bg=(0.0, 0.0, 0.0)),
    # BASELINE configs[3]: Frosting refine step, 2 M shell-bound Gaussians + 200 704-triangle occlusion mesh
    "c4": dict(P=2_000_000, width=1600, height=1056, fx=1334.0, fy=1334.0, seed=SEED_BASE + 4, bg=(0.0, 0.0, 0.0),
               kind="shell", n_lat=224, n_lon=448),
}


def sphere_mesh(n_lat: int, n_lon: int, radius: float = 1.0):
    """Lat-long sphere with 2 * n_lat * n_lon triangles (the C4 shell stand-in):
    verts [ (n_lat + 1) * n_lon, 3 ] float32, faces [F, 3] int32."""
    th = torch.linspace(0, math.pi, n_lat + 1, dtype=torch.float64)
    ph = torch.linspace(0, 2 * math.pi, n_lon + 1, dtype=torch.float64)[:-1]
    T, Pp = torch.meshgrid(th, ph, indexing="ij")
    verts = (torch.stack([torch.sin(T) * torch.cos(Pp), torch.cos(T), torch.sin(T) * torch.sin(Pp)], -1).reshape(-1, 3) * radius).float()
    i, j = torch.meshgrid(torch.arange(n_lat), torch.arange(n_lon), indexing="ij")
    a, b = (i * n_lon + j).reshape(-1), (i * n_lon + (j + 1) % n_lon).reshape(-1)
    c, d = ((i + 1) * n_lon + j).reshape(-1), ((i + 1) * n_lon + (j + 1) % n_lon).reshape(-1)
    faces = torch.cat([torch.stack([a, c, b], 1), torch.stack([b, c, d], 1)]).int()
    return verts.contiguous(), faces.contiguous()


@dataclass
class ShellScene:
    """BASELINE configs[3] (C4): Gaussians bound to the cells of a shell mesh + that mesh for occlusion culling."""
    scene: Scene
    verts: torch.Tensor      # [V,3] float32 (the shell's base mesh, frosting_model.py:1534-1535)
    faces: torch.Tensor      # [F,3] int32
    cell: torch.Tensor       # [P] int64: base face of each Gaussian's cell (_point_cell_indices)

    def to(self, device):
        return ShellScene(self.scene.to(device), self.verts.to(device), self.faces.to(device), self.cell.to(device))


def make_shell_scene(P: int, seed: int, n_lat: int = 224, n_lon: int = 448) -> ShellScene:
    """SURVEY.md 8(d) C4 generator: lat-long unit sphere (224 x 448 -> 200 704 triangles); Gaussians =
    area-weighted face pick + Dirichlet(1,1,1) barycentrics + normal offset U(-0.02, 0.02)."""
    verts, faces = sphere_mesh(n_lat, n_lon)
    g = torch.Generator().manual_seed(seed)
    v = verts[faces.long()].double()
    area = torch.linalg.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]).norm(dim=1)
    cell = torch.multinomial(area / area.sum(), P, replacement=True, generator=g)
    e = -torch.log(torch.rand(P, 3, generator=g, dtype=torch.float64).clamp_min(1e-300))   # Dirichlet(1,1,1)
    bary = e / e.sum(1, keepdim=True)
    pts = (v[cell] * bary[:, :, None]).sum(1)
    nrm = pts / pts.norm(dim=1, keepdim=True).clamp_min(1e-12)
    pts = pts + nrm * (torch.rand(P, 1, generator=g, dtype=torch.float64) * 0.04 - 0.02)
    sc = make_scene(P, seed + 1000)
    scene = Scene(pts.float().contiguous(), sc.scales, sc.rotations, sc.opacities, sc.shs, sc.sh_degree)
    return ShellScene(scene, verts, faces, cell.contiguous())


def make_skew_scene(P: int, seed: int, centres=None, cluster_sigma: float = 0.03, cluster_frac: float = 0.5,
                    n_big: int | None = None, big_scale: float = 0.05) -> Scene:
    """Stress scene for the binning / sort / blend balance: a fraction of the Gaussians in a few
    tight clusters (tile lists of 10^4-10^5 entries next to empty tiles), a few hundred large near-camera
    Gaussians (tile rectangles of hundreds of tiles), the rest a thin uniform ball.  The defaults are the scene
    bench.py reports as `skew_scene` (unchanged since round 2); the keyword arguments shape the small-image variants
    of the parity tests."""
    g = torch.Generator().manual_seed(seed)
    base = make_scene(P, seed + 1)
    f64 = torch.float64
    n_cl = int(P * cluster_frac)
    if centres is None:
        centres = [[0.0, 0.0, 0.0], [0.9, 0.3, 0.2], [-0.8, -0.4, 0.5], [0.2, 0.7, -0.6]]
    centres = torch.tensor(centres, dtype=f64)
    which = torch.randint(0, centres.shape[0], (n_cl,), generator=g)
    means = base.means3D.double().clone()
    means[:n_cl] = centres[which] + cluster_sigma * torch.randn(n_cl, 3, generator=g, dtype=f64)
    scales = base.scales.double().clone()
    n_big = min(400, P // 100) if n_big is None else n_big
    if n_big:
        # large and close to camera 0 (which sits at (0, 0, -4) looking at the origin)
        means[n_cl:n_cl + n_big] = torch.tensor([0.0, 0.0, -2.8], dtype=f64) + torch.tensor([0.8, 0.5, 0.3], dtype=f64) * \
            torch.randn(n_big, 3, generator=g, dtype=f64)
        scales[n_cl:n_cl + n_big] = big_scale * torch.exp(0.5 * torch.randn(n_big, 3, generator=g, dtype=f64))
    return Scene(means.float().contiguous(), scales.float().contiguous(), base.rotations, base.opacities, base.shs,
                 base.sh_degree)


def long_list_scene(P: int = 2_000_000, seed: int = SEED_BASE + 31):
    """(Scene, Camera, bg) of the small-image stress case of the parity tests: 320x240 (300 tiles), two very tight
    clusters (one tile list of more than 250 000 entries each), a dense ball behind them (more than a hundred lists
    beyond the 8192-entry LDS capacity of the tile sort) and 200 Gaussians that each cover the whole image (their
    64-Gaussian waves own ~19 000 backward slots)."""
    scene = make_skew_scene(P, seed, centres=[[0.0, 0.0, 0.0], [0.5, 0.3, 0.2]], cluster_sigma=0.01, cluster_frac=0.3,
                            n_big=200, big_scale=0.3)
    cam = ring_camera(0, 320, 240, 267.0, 267.0)
    return scene, cam, torch.zeros(3)


def config_scene(name: str, view: int = 0, P: int | None = None):
    cfg = CONFIGS[name]
    if cfg.get("kind") == "shell":
        raise ValueError(f"{name} is a shell config: use config_shell_scene()")
    scene = make_scene(P or cfg["P"], cfg["seed"], log_scale=cfg.get("log_scale", math.log(0.005)))
    cam = ring_camera(view, cfg["width"], cfg["height"], cfg["fx"], cfg["fy"])
    bg = torch.tensor(cfg["bg"], dtype=torch.float32)
    return scene, cam, bg


def l1_target_grad(image: torch.Tensor, seed: int):
    """dL/dimage for loss = mean|image - target| with a seeded random target
    (same L1 as frosting_utils/loss_utils.py:17-18)."""
    g = torch.Generator().manual_seed(seed)
    target = torch.rand(image.shape, generator=g).to(image.device)
    return torch.sign(image - target) / image.numel(), target


def config_shell_scene(name: str = "c4", view: int = 0, P: int | None = None, n_lat: int | None = None,
                       n_lon: int | None = None):
    """(ShellScene, Camera, bg) of a shell config (C4)."""
    cfg = CONFIGS[name]
    shell = make_shell_scene(P or cfg["P"], cfg["seed"], n_lat or cfg["n_lat"], n_lon or cfg["n_lon"])
    cam = ring_camera(view, cfg["width"], cfg["height"], cfg["fx"], cfg["fy"])
    return shell, cam, torch.tensor(cfg["bg"], dtype=torch.float32)

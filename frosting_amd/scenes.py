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
                   znear: float = 0.01, zfar: float = 100.0) -> Camera:
    """COLMAP axes (x right, y down, z forward), looking at the origin."""
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
    world_view = wv64.float()
    full = (wv64 @ proj64).float()
    campos = c.float().contiguous()
    return Camera(height, width, float(tanfovx), float(tanfovy), world_view, full.contiguous(), campos)


def ring_camera(k: int, width: int, height: int, fx: float, fy: float, n: int = 8, dist: float = 4.0) -> Camera:
    th = 2.0 * math.pi * k / n
    return look_at_camera([dist * math.sin(th), 0.0, -dist * math.cos(th)], width, height, fx, fy)


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
}


def config_scene(name: str, view: int = 0, P: int | None = None):
    cfg = CONFIGS[name]
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

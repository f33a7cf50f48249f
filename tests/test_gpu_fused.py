"""Raw-parameter mode: activations and Frosting's shell-bound centres evaluated inside the
per-Gaussian kernels, against the reference's formulation -- the eager torch chain of
frosting_scene/frosting_model.py:707-798 in front of the rasterizer, differentiated by autograd."""
import pytest
import torch
import torch.nn.functional as F

from frosting_amd import scenes
from frosting_amd.fused import rasterize_raw

import helpers as Hh

pytestmark = pytest.mark.gpu


def _raw_model(scene, dev, seed):
    """Raw parameters whose activations reproduce `scene` (inverse sigmoid / log / unnormalised quaternion)."""
    g = torch.Generator().manual_seed(seed)
    o = scene.opacities.double().clamp(1e-4, 1 - 1e-4)
    raw_o = torch.log(o / (1 - o)).float().reshape(-1)
    raw_s = torch.log(scene.scales.double()).float()
    raw_r = (scene.rotations.double() * (0.5 + torch.rand(scene.P, 1, generator=g, dtype=torch.float64))).float()
    return [t.to(dev).requires_grad_(True) for t in (raw_o, raw_s, raw_r)]


def _reference_chain(settings, shs, raw_o, raw_s, raw_r, means):
    from diff_gaussian_rasterization import GaussianRasterizer
    rast = GaussianRasterizer(settings)
    return rast(means3D=means, means2D=torch.zeros_like(means), shs=shs, colors_precomp=None,
                opacities=torch.sigmoid(raw_o).view(-1, 1), scales=torch.exp(raw_s), rotations=F.normalize(raw_r, dim=-1),
                cov3D_precomp=None)


def _ulp_noise(run_chain, leaves, names, gpix, seeds=(1, 2)):
    """How far the reference's OWN formulation (torch activation chain + rasterizer + autograd) moves when its raw
    parameters move by one float32 ulp in a random direction: the two paths compared below feed the rasterizer inputs
    that differ by exactly such ulps (our sigmoid / exp / softmax are not bit-identical to torch's), and the scale /
    quaternion gradients are sums with heavy cancellation that amplify them.  Largest rel. L2 over the seeds, per leaf."""
    def grads(ls):
        img, _ = run_chain(ls)
        img.backward(gpix)
        return [l.grad.clone() for l in ls]
    base = grads([l.detach().clone().requires_grad_(True) for l in leaves])
    noise = {n: 0.0 for n in names}
    for seed in seeds:
        g = torch.Generator().manual_seed(seed)
        moved = []
        for l in leaves:
            d = l.detach()
            up = (torch.rand(d.shape, generator=g) < 0.5).to(d.device)
            inf = torch.full_like(d, float("inf"))
            moved.append(torch.where(up, torch.nextafter(d, inf), torch.nextafter(d, -inf)).requires_grad_(True))
        for n, g0, g1 in zip(names, base, grads(moved)):
            noise[n] = max(noise[n], Hh.rel_l2(g1, g0))
    return noise


# floors: ~3 x the largest difference measured where the ulp noise is below it (profiles/r02_pytest_gpu.log)
FLOOR = {"shs": 1e-6, "raw_opacity": 1.5e-6, "means3D": 1.6e-5, "logits": 1e-5, "cell_verts": 1e-5, "raw_scale": 1.5e-4,
         "raw_rot": 3e-4}


def _compare(img_a, radii_a, leaves_a, img_b, radii_b, leaves_b, names, gpix, noise):
    assert float((img_a.detach() - img_b.detach()).abs().mean()) < 2e-7                    # activations differ by an ulp here and there
    assert float((radii_a != radii_b).float().mean()) < 1e-4
    img_a.backward(gpix)
    img_b.backward(gpix)
    errs = {n: Hh.rel_l2(a.grad, b.grad) for n, a, b in zip(names, leaves_a, leaves_b)}
    print("\nrel-L2 of the gradients, fused vs torch chain (the chain against itself, inputs moved by one ulp): " +
          ", ".join(f"{k} {v:.1e} ({noise[k]:.1e})" for k, v in errs.items()))
    for n, a in zip(names, leaves_a):
        assert errs[n] < max(5.0 * noise[n], FLOOR[n]), (n, errs[n], noise[n])
        assert bool(torch.isfinite(a.grad).all())


def test_raw_parameters_equal_the_torch_activation_chain(gpu_device):
    dev = gpu_device
    scene, cam, bg = scenes.config_scene("c2", 2, P=30_000)
    settings = Hh.settings_for(cam, bg, 3, dev)
    names = ["shs", "raw_opacity", "raw_scale", "raw_rot", "means3D"]
    leaves = []
    for _ in range(2):
        raw = _raw_model(scene, dev, 11)
        leaves.append([scene.shs.to(dev).requires_grad_(True)] + raw + [scene.means3D.to(dev).requires_grad_(True)])
    a, b = leaves
    img_a, radii_a = rasterize_raw(settings, a[0], a[1], a[2], a[3], means3D=a[4])
    img_b, radii_b = _reference_chain(settings, b[0], b[1], b[2], b[3], b[4])
    gpix, _ = scenes.l1_target_grad(img_b.detach().cpu(), 3)
    gpix = gpix.to(dev)
    noise = _ulp_noise(lambda ls: _reference_chain(settings, *ls), b, names, gpix)
    _compare(img_a, radii_a, a, img_b, radii_b, b, names, gpix, noise)


def test_means2D_receives_the_viewspace_gradient_and_inplace_updates_are_caught(gpu_device):
    """the raw path returns the reference's viewspace gradient through an optional means2D input
    (gaussian_model.py:404-407 reads it for densification), and its inputs are saved with save_for_backward, so an
    in-place optimizer step between forward and backward raises instead of silently using the new values."""
    dev = gpu_device
    scene, cam, bg = scenes.config_scene("c2", 3, P=20_000)
    settings = Hh.settings_for(cam, bg, 3, dev)
    raw = _raw_model(scene, dev, 17)
    shs, means = scene.shs.to(dev).requires_grad_(True), scene.means3D.to(dev).requires_grad_(True)
    means2D = torch.zeros_like(means, requires_grad=True)
    img, radii = rasterize_raw(settings, shs, raw[0], raw[1], raw[2], means3D=means, means2D=means2D)
    gpix, _ = scenes.l1_target_grad(img.detach().cpu(), 4)
    img.backward(gpix.to(dev), retain_graph=True)
    assert means2D.grad is not None and float(means2D.grad.abs().sum()) > 0 and not means2D.grad[:, 2].any()
    # the same screen-space gradient as the drop-in rasterizer's means2D input
    from diff_gaussian_rasterization import GaussianRasterizer
    m2 = torch.zeros_like(means, requires_grad=True)
    img_c, _ = GaussianRasterizer(settings)(means3D=means.detach(), means2D=m2, shs=shs.detach(), colors_precomp=None,
                                            opacities=torch.sigmoid(raw[0].detach()).view(-1, 1), scales=torch.exp(raw[1].detach()),
                                            rotations=F.normalize(raw[2].detach(), dim=-1), cov3D_precomp=None)
    img_c.backward(gpix.to(dev))
    assert Hh.rel_l2(means2D.grad, m2.grad) < 1e-4
    with torch.no_grad():
        raw[1].add_(0.01)                     # an optimizer step in place
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        img.backward(gpix.to(dev))


@pytest.mark.parametrize("learn_shell,softmax", [(False, True), (True, True), (True, False)])
def test_shell_bound_centres_and_learnable_shell(gpu_device, learn_shell, softmax):
    """Frosting's parameterisation: centres = barycentric weights . prism vertices (frosting_model.py:707-724), the
    weights softmax(logits) or -- use_softmax_for_bary_coords = False, :716-718 -- relu(x) / sum relu(x); gradients
    w.r.t. the logits and, with learn_shell = True, w.r.t. the cell vertices (several Gaussians per cell)."""
    dev = gpu_device
    P = 40_000
    shell, cam, bg = scenes.config_shell_scene("c4", 1, P=P, n_lat=40, n_lon=80)     # 6 400 cells, ~6 Gaussians each
    cam = scenes.ring_camera(1, 800, 528, 667.0, 667.0)
    settings = Hh.settings_for(cam, bg, 3, dev)
    verts, faces = shell.verts.double(), shell.faces.long()
    nrm = verts / verts.norm(dim=1, keepdim=True)
    inner, outer = (verts - 0.03 * nrm), (verts + 0.03 * nrm)
    cell_verts = torch.stack([inner[faces], outer[faces]], dim=1).float()           # [F,2,3,3] = shell_cells_verts
    g = torch.Generator().manual_seed(5)
    logits0 = torch.randn(P, 6, generator=g)
    if not softmax:     # the reference initialises non-softmax coordinates as barycentric weights (:502-511); some negative ones
        logits0 = torch.rand(P, 6, generator=g) - 0.15    # exercise the relu (and its zero gradient)
        logits0[:, 0] = logits0[:, 0].abs() + 0.1         # (never an all-negative row: the reference divides by the sum)
    sc = shell.scene
    names = ["shs", "raw_opacity", "raw_scale", "raw_rot", "logits"] + (["cell_verts"] if learn_shell else [])
    leaves = []
    for _ in range(2):
        raw = _raw_model(sc, dev, 13)
        ls = [sc.shs.to(dev).requires_grad_(True)] + raw + [logits0.to(dev).requires_grad_(True)]
        cv = cell_verts.to(dev).requires_grad_(learn_shell)
        leaves.append((ls + ([cv] if learn_shell else []), cv))
    (a, cv_a), (b, cv_b) = leaves
    cells = shell.cell.to(dev)
    img_a, radii_a = rasterize_raw(settings, a[0], a[1], a[2], a[3], shell_logits=a[4], shell_cell_verts=cv_a, shell_cells=cells,
                                   use_softmax_for_bary_coords=softmax)

    def bary(x):                                         # frosting_model.py:713-719
        if softmax:
            return torch.softmax(x, dim=-1)
        r = torch.relu(x)
        return r / r.sum(dim=-1, keepdim=True)

    def chain(ls):
        cv = ls[5] if learn_shell else cv_b
        means = (bary(ls[4])[..., None] * cv[cells].reshape(-1, 6, 3)).sum(dim=-2)   # the reference's `points`
        return _reference_chain(settings, ls[0], ls[1], ls[2], ls[3], means)
    img_b, radii_b = chain(b)
    assert int((radii_b > 0).sum()) > 5000
    gpix, _ = scenes.l1_target_grad(img_b.detach().cpu(), 7)
    gpix = gpix.to(dev)
    noise = _ulp_noise(chain, b, names, gpix)
    _compare(img_a, radii_a, a, img_b, radii_b, b, names, gpix, noise)
    if not softmax:
        assert bool((a[4].grad[a[4].detach() <= 0] == 0).all())      # relu: no gradient where the coordinate is clipped
    if learn_shell:
        assert float(cv_a.grad.abs().sum()) > 0 and cv_a.grad.shape == cell_verts.shape


def test_argument_rules(gpu_device):
    dev = gpu_device
    scene, cam, bg = scenes.config_scene("mini", 0, P=200)
    settings = Hh.settings_for(cam, bg, 3, dev)
    sc = scene.to(dev)
    with pytest.raises(Exception, match="exactly one"):
        rasterize_raw(settings, sc.shs, sc.opacities, sc.scales, sc.rotations)
    with pytest.raises(RuntimeError, match="shell_cell_verts"):
        rasterize_raw(settings, sc.shs, sc.opacities, sc.scales, sc.rotations, shell_logits=torch.zeros(200, 6, device=dev))

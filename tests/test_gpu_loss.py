"""Fused photometric loss against the reference-generated vectors and the
torch restatement of frosting_utils/loss_utils.py."""
import os

import numpy as np
import pytest
import torch

import torch_ref as TR
from helpers import rel_l2
from frosting_amd.loss import photometric_loss, photometric_loss_and_grad

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_l1_dssim.npz")


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_against_reference_vectors(gpu_device, name):
    d = np.load(GOLD)
    pred = torch.from_numpy(d[f"{name}_pred"]).to(gpu_device)
    gt = torch.from_numpy(d[f"{name}_gt"]).to(gpu_device)
    loss, grad = photometric_loss_and_grad(pred, gt)
    assert abs(float(loss) - float(d[f"{name}_loss"])) <= 1e-6
    assert rel_l2(grad.cpu(), d[f"{name}_grad"]) <= 2e-5
    np.testing.assert_allclose(grad.cpu().numpy(), d[f"{name}_grad"], rtol=2e-4, atol=2e-9)
    loss2, grad2 = photometric_loss_and_grad(pred, gt)          # fixed-order reduction
    assert torch.equal(loss, loss2) and torch.equal(grad, grad2)


@pytest.mark.parametrize("shape,lam", [((3, 112, 160), 0.2), ((3, 1056, 1600), 0.2), ((3, 33, 17), 0.5), ((1, 16, 16), 0.0)])
def test_against_torch_restatement(gpu_device, shape, lam):
    g = torch.Generator().manual_seed(shape[1])
    gt = torch.rand(shape, generator=g)
    pred0 = (gt + 0.2 * torch.randn(shape, generator=g)).clamp(0, 1)
    pred_r = pred0.clone().to(gpu_device).requires_grad_(True)
    ref = TR.photometric_loss_ref(pred_r, gt.to(gpu_device), lam)
    ref.backward()
    pred_o = pred0.clone().to(gpu_device).requires_grad_(True)
    ours = photometric_loss(pred_o, gt.to(gpu_device), lam)       # autograd wrapper
    (3.0 * ours).backward()
    assert abs(ours.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    assert rel_l2(pred_o.grad.cpu(), 3.0 * pred_r.grad.cpu()) <= 5e-5
    value_only, none = photometric_loss_and_grad(pred0.to(gpu_device), gt.to(gpu_device), lam, need_grad=False)
    assert none is None and value_only.item() == ours.item()


def test_gradient_feeds_the_rasterizer_backward(gpu_device):
    """loss(render) end to end through the drop-in rasterizer: same parameter gradients as with the
    torch restatement of the loss."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from frosting_amd import scenes
    from helpers import settings_for
    dev = gpu_device
    scene, cam, bg = scenes.config_scene("mini", 1, P=2000)
    sc = scene.to(dev)
    target = torch.rand(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(1)).to(dev)
    rast = GaussianRasterizer(settings_for(cam, bg, scene.sh_degree, dev))
    grads = []
    for loss_fn in (photometric_loss, TR.photometric_loss_ref):
        p = {n: getattr(sc, n).clone().requires_grad_(True) for n in ("means3D", "opacities", "shs", "scales", "rotations")}
        img, _ = rast(means3D=p["means3D"], means2D=torch.zeros_like(p["means3D"], requires_grad=True), opacities=p["opacities"],
                      shs=p["shs"], scales=p["scales"], rotations=p["rotations"])
        loss_fn(img, target).backward()
        grads.append({n: t.grad.clone() for n, t in p.items()})
    for n in grads[0]:
        assert rel_l2(grads[0][n], grads[1][n]) <= 2e-4, n

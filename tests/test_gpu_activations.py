"""Fused parameter activations against torch's sigmoid / exp / F.normalize and their autograd."""
import pytest
import torch

from frosting_amd.activations import activate, activate_backward_, gaussian_activations

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [1, 1000, 4097])
def test_activations_match_torch(gpu_device, P):
    dev = gpu_device
    g = torch.Generator().manual_seed(P)
    ro = (4 * torch.randn(P, 1, generator=g)).to(dev)
    rs = (torch.randn(P, 3, generator=g) - 4).to(dev)
    rr = torch.randn(P, 4, generator=g).to(dev)
    rr[0] = 0.0                                             # degenerate quaternion: F.normalize's eps branch
    a = [t.clone().requires_grad_(True) for t in (ro, rs, rr)]
    b = [t.clone().requires_grad_(True) for t in (ro, rs, rr)]
    ours = gaussian_activations(*a)
    ref = (torch.sigmoid(b[0]), torch.exp(b[1]), torch.nn.functional.normalize(b[2]))
    w = [torch.randn(t.shape, generator=g).to(dev) for t in ref]
    sum((x * y).sum() for x, y in zip(ours, w)).backward()
    sum((x * y).sum() for x, y in zip(ref, w)).backward()
    for x, y in zip(ours, ref):
        torch.testing.assert_close(x, y.detach(), rtol=2e-6, atol=1e-7)
    for x, y in zip(a, b):
        torch.testing.assert_close(x.grad, y.grad, rtol=1e-5, atol=1e-6 * float(y.grad.abs().max()) + 1e-12)


def test_in_place_backward_on_a_flat_gradient_buffer(gpu_device):
    dev = gpu_device
    P = 77
    g = torch.Generator().manual_seed(0)
    ro, rs, rr = torch.randn(P, 1, generator=g).to(dev), torch.randn(P, 3, generator=g).to(dev), torch.randn(P, 4, generator=g).to(dev)
    o, s, r = activate(ro, rs, rr)
    flat = torch.randn(P * 8, generator=g).to(dev)          # [scales | rotations | opacities] like the exchange buffer
    gs, gr, go = flat[: 3 * P].view(P, 3), flat[3 * P: 7 * P].view(P, 4), flat[7 * P:].view(P, 1)
    want_o = go * o * (1 - o)
    want_s = gs * s
    activate_backward_(o, s, rr, go, gs, gr)
    torch.testing.assert_close(go, want_o, rtol=2e-6, atol=1e-9)
    torch.testing.assert_close(gs, want_s, rtol=2e-6, atol=1e-9)
    assert float((gr * r).sum(dim=1).abs().max()) < 1e-5    # the normalisation's backward is orthogonal to the unit quaternion
    with pytest.raises(RuntimeError, match="GPU only"):
        activate(ro.cpu(), rs.cpu(), rr.cpu())


def test_shell_points_match_the_reference_formula(gpu_device):
    """frosting_model.py:713-724 in torch (softmax, gather, weighted sum) against the fused kernels."""
    from frosting_amd.activations import shell_points
    dev = gpu_device
    g = torch.Generator().manual_seed(2)
    F, P = 500, 7001
    cells = torch.randn(F, 2, 3, 3, generator=g).to(dev)                       # shell_cells_verts
    idx = torch.randint(0, F, (P,), generator=g).to(dev)
    lg0 = (3 * torch.randn(P, 6, generator=g)).to(dev)
    a = lg0.clone().requires_grad_(True)
    b = lg0.clone().requires_grad_(True)
    ours = shell_points(a, cells, idx)
    ref = (torch.softmax(b, dim=-1)[..., None] * cells[idx].reshape(-1, 6, 3)).sum(dim=-2)
    torch.testing.assert_close(ours, ref.detach(), rtol=2e-6, atol=2e-6)
    w = torch.randn(P, 3, generator=g).to(dev)
    (ours * w).sum().backward()
    (ref * w).sum().backward()
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-5, atol=2e-6)
    with pytest.raises(RuntimeError, match="int64"):
        shell_points(a, cells, idx.int())

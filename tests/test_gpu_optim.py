"""Fused Adam against torch.optim.Adam configured as the reference configures it
(frosting_optimizer.py:74-121: one group per tensor, lr per group, eps=1e-15)."""
import pytest
import torch

from frosting_amd.optim import FlatAdam
from frosting_amd.parallel import PARAM_ORDER

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,K", [(1000, 16), (257, 4), (3, 1)])
def test_flat_adam_matches_torch_adam(gpu_device, P, K):
    dev = gpu_device
    shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
    lrs = dict(means3D=1.6e-4, scales=5e-3, rotations=1e-3, opacities=5e-2, shs=2.5e-3)
    g = torch.Generator().manual_seed(P)
    init = {k: torch.randn(shapes[k], generator=g) for k in PARAM_ORDER}
    ours = FlatAdam(shapes, lrs, dev)
    ref_params = []
    for k in PARAM_ORDER:
        ours.params[k].copy_(init[k])
        ref_params.append(torch.nn.Parameter(init[k].clone().to(dev)))
    ref = torch.optim.Adam([{"params": [p], "lr": lrs[k]} for p, k in zip(ref_params, PARAM_ORDER)], lr=0.0, eps=1e-15)
    for step in range(12):
        grads = {k: (torch.randn(shapes[k], generator=g) * 10.0 ** float(torch.randint(-6, 1, (1,), generator=g))) for k in PARAM_ORDER}
        if step == 5:
            grads["opacities"].zero_()                      # a group without gradient signal this step
            ours.set_lr("means3D", 0.8e-4)                  # schedule hook
            ref.param_groups[0]["lr"] = 0.8e-4
        flat = torch.cat([grads[k].reshape(-1) for k in PARAM_ORDER]).to(dev)
        ours.step(flat)
        for p, k in zip(ref_params, PARAM_ORDER):
            p.grad = grads[k].to(dev)
        ref.step()
    def close(a, b):
        # a few float32 roundings of the largest magnitude in play (ATen may contract a*b+c, we do not);
        # entries that cancelled down to much smaller values carry that absolute error too
        torch.testing.assert_close(a, b, rtol=2e-6, atol=4e-7 * float(b.abs().max()))

    for p, k in zip(ref_params, PARAM_ORDER):
        close(ours.params[k], p.detach())
        st = ref.state[p]
        lo = sum(int(torch.Size(shapes[n]).numel()) for n in PARAM_ORDER[:PARAM_ORDER.index(k)])
        hi = lo + p.numel()
        close(ours.exp_avg[lo:hi].view_as(p), st["exp_avg"])
        close(ours.exp_avg_sq[lo:hi].view_as(p), st["exp_avg_sq"])


def test_flat_adam_grad_scale_and_errors(gpu_device):
    dev = gpu_device
    shapes = dict(means3D=(10, 3), opacities=(10, 1))
    a = FlatAdam(shapes, dict(means3D=1e-2, opacities=1e-2), dev)
    b = FlatAdam(shapes, dict(means3D=1e-2, opacities=1e-2), dev)
    g = torch.randn(40, device=dev)
    a.step(g, grad_scale=0.125)          # mean over 8 views folded into the update
    b.step(g * 0.125)
    assert torch.equal(a.flat, b.flat)
    with pytest.raises(RuntimeError, match="gradient buffer"):
        a.step(g[:-1])
    with pytest.raises(RuntimeError, match="GPU only"):
        a.step(g.cpu())

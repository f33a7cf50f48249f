"""Golden vectors of the photometric loss from the REFERENCE's own Python (imported from
/root/reference in this container; it cannot travel to the GPU box, the fixture does):
tests/golden/loss_l1_dssim.npz = inputs, loss value and d loss / d pred from autograd on CPU."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from frosting_utils.loss_utils import ssim, l1_loss   # noqa: E402

g = torch.Generator().manual_seed(20241022)
out = {}
for name, (C, H, W) in dict(a=(3, 40, 56), b=(3, 23, 17), c=(1, 12, 12)).items():
    gt = torch.rand(C, H, W, generator=g)
    pred = (gt + 0.15 * torch.randn(C, H, W, generator=g)).clamp(0, 1.2)
    pred[:, :3, :4] = gt[:, :3, :4]                      # exact matches: sign(0) = 0 in the L1 term
    pred.requires_grad_(True)
    loss = 0.8 * l1_loss(pred[None], gt[None]) + 0.2 * (1.0 - ssim(pred[None], gt[None]))
    loss.backward()
    out[f"{name}_pred"] = pred.detach().numpy(); out[f"{name}_gt"] = gt.numpy()
    out[f"{name}_loss"] = np.float32(loss.item()); out[f"{name}_grad"] = pred.grad.numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "loss_l1_dssim.npz"), **out)
print({k: v.shape for k, v in out.items()})

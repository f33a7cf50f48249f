"""The torch restatement of the reference's photometric loss (tests/torch_ref.py) is pinned on vectors
produced by the reference's own loss_utils.py (tools/make_golden_loss.py)."""
import os

import numpy as np
import pytest
import torch

import torch_ref as TR
from frosting_amd.loss import gaussian_window

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_l1_dssim.npz")


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_restatement_matches_reference_vectors(name):
    d = np.load(GOLD)
    pred = torch.from_numpy(d[f"{name}_pred"]).clone().requires_grad_(True)
    gt = torch.from_numpy(d[f"{name}_gt"])
    loss = TR.photometric_loss_ref(pred, gt)
    loss.backward()
    assert abs(loss.item() - float(d[f"{name}_loss"])) <= 2e-7
    np.testing.assert_allclose(pred.grad.numpy(), d[f"{name}_grad"], rtol=1e-5, atol=1e-9)


def test_window_is_the_references():
    w = gaussian_window()
    assert w.dtype == torch.float32 and w.shape == (11,)
    assert abs(float(w.sum()) - 1.0) < 1e-6 and torch.equal(w, w.flip(0))
    assert abs(float(w[5]) - 0.26601) < 1e-4


def test_loss_has_no_cpu_path():
    from frosting_amd.loss import photometric_loss
    with pytest.raises(RuntimeError, match="GPU only"):
        photometric_loss(torch.zeros(3, 8, 8), torch.zeros(3, 8, 8))

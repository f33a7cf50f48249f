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
        flat = torch.zeros(ours.numel, device=dev)          # the optimizer's layout (16-byte aligned segments)
        for k, (o, n) in ours.layout.items():
            flat[o:o + n] = grads[k].reshape(-1).to(dev)
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
        close(ours.m[k], st["exp_avg"])
        close(ours.v[k], st["exp_avg_sq"])


def test_flat_adam_grad_scale_and_errors(gpu_device):
    dev = gpu_device
    shapes = dict(means3D=(10, 3), opacities=(10, 1))
    a = FlatAdam(shapes, dict(means3D=1e-2, opacities=1e-2), dev)
    b = FlatAdam(shapes, dict(means3D=1e-2, opacities=1e-2), dev)
    g = torch.randn(a.numel, device=dev)
    a.step(g, grad_scale=0.125)          # mean over 8 views folded into the update
    b.step(g * 0.125)
    assert torch.equal(a.flat, b.flat)
    with pytest.raises(RuntimeError, match="gradient buffer"):
        a.step(g[:-1])
    with pytest.raises(RuntimeError, match="GPU only"):
        a.step(g.cpu())


def test_flat_adam_dc_and_rest_groups_on_one_sh_tensor(gpu_device):
    """sh_dc_lr: one [P,16,3] SH tensor stepped like the reference's two groups (features_dc at lr,
    features_rest at lr / 20) -- no torch.cat, no gradient split."""
    dev = gpu_device
    P, K = 333, 16
    shapes = dict(means3D=(P, 3), shs=(P, K, 3))
    g = torch.Generator().manual_seed(3)
    sh0, m0 = torch.randn(P, K, 3, generator=g), torch.randn(P, 3, generator=g)
    ours = FlatAdam(shapes, dict(means3D=1e-3, shs=2.5e-3 / 20.0), dev, sh_dc_lr=2.5e-3)
    ours.params["means3D"].copy_(m0); ours.params["shs"].copy_(sh0)
    dc = torch.nn.Parameter(sh0[:, :1].clone().to(dev)); rest = torch.nn.Parameter(sh0[:, 1:].clone().to(dev))
    xyz = torch.nn.Parameter(m0.clone().to(dev))
    ref = torch.optim.Adam([{"params": [xyz], "lr": 1e-3}, {"params": [dc], "lr": 2.5e-3},
                            {"params": [rest], "lr": 2.5e-3 / 20.0}], lr=0.0, eps=1e-15)
    for _ in range(6):
        gs, gm = torch.randn(P, K, 3, generator=g) * 1e-2, torch.randn(P, 3, generator=g) * 1e-2
        flat = torch.zeros(ours.numel, device=dev)
        for k, t in (("means3D", gm), ("shs", gs)):
            o, n = ours.layout[k]
            flat[o:o + n] = t.reshape(-1).to(dev)
        ours.step(flat)
        xyz.grad, dc.grad, rest.grad = gm.to(dev), gs[:, :1].contiguous().to(dev), gs[:, 1:].contiguous().to(dev)
        ref.step()
    want = torch.cat([dc.detach(), rest.detach()], dim=1)
    torch.testing.assert_close(ours.params["shs"], want, rtol=2e-6, atol=4e-7 * float(want.abs().max()))
    torch.testing.assert_close(ours.params["means3D"], xyz.detach(), rtol=2e-6, atol=4e-7 * float(xyz.detach().abs().max()))
    # the two learning rates really differ: DC moved ~20x further than the rest
    moved = (ours.params["shs"].cpu() - sh0).abs()
    assert float(moved[:, 0].mean()) > 10 * float(moved[:, 1:].mean())


def test_flat_adam_prune_append_reset_follow_torch(gpu_device):
    """Densification / pruning (gaussian_model.py: _prune_optimizer, cat_tensors_to_optimizer,
    replace_tensor_to_optimizer): the surviving rows keep parameters and both moments, new rows start
    with zero moments -- checked against torch.optim.Adam whose state is edited the reference's way."""
    dev = gpu_device
    P, K = 37, 4
    shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
    lrs = dict(means3D=1e-3, scales=5e-3, rotations=1e-3, opacities=5e-2, shs=2.5e-3)
    g = torch.Generator().manual_seed(7)
    ours = FlatAdam(shapes, lrs, dev)
    ref_p = {}
    for k in PARAM_ORDER:
        init = torch.randn(shapes[k], generator=g)
        ours.params[k].copy_(init)
        ref_p[k] = torch.nn.Parameter(init.clone().to(dev))
    ref = torch.optim.Adam([{"params": [ref_p[k]], "lr": lrs[k], "name": k} for k in PARAM_ORDER], lr=0.0, eps=1e-15)

    def both_step():
        flat = torch.zeros(ours.numel, device=dev)
        for k, (o, n) in ours.layout.items():
            gr = torch.randn(ours.shapes[k], generator=g).to(dev)
            flat[o:o + n] = gr.reshape(-1)
            ref_p[k].grad = gr
        ours.step(flat)
        ref.step()

    def ref_replace(k, tensor, m, v):
        grp = next(gr for gr in ref.param_groups if gr["name"] == k)
        st = ref.state.pop(grp["params"][0])
        newp = torch.nn.Parameter(tensor)
        st["exp_avg"], st["exp_avg_sq"] = m, v
        grp["params"][0] = newp
        ref.state[newp] = st
        ref_p[k] = newp

    for _ in range(3):
        both_step()
    keep = torch.rand(P, generator=g) > 0.3
    ours.prune(keep)
    kd = keep.to(dev)
    for k in PARAM_ORDER:
        st = ref.state[ref_p[k]]
        ref_replace(k, ref_p[k].detach()[kd].clone(), st["exp_avg"][kd].clone(), st["exp_avg_sq"][kd].clone())
    n_new = 9
    new = {k: torch.randn((n_new,) + tuple(shapes[k][1:]), generator=g) for k in PARAM_ORDER}
    ours.append(new)
    for k in PARAM_ORDER:
        st = ref.state[ref_p[k]]
        z = torch.zeros((n_new,) + tuple(shapes[k][1:]), device=dev)
        ref_replace(k, torch.cat([ref_p[k].detach(), new[k].to(dev)]), torch.cat([st["exp_avg"], z]), torch.cat([st["exp_avg_sq"], z]))
    ours.reset("opacities", torch.full_like(ours.params["opacities"], 0.01))
    ref_replace("opacities", torch.full_like(ref_p["opacities"].detach(), 0.01), torch.zeros_like(ref_p["opacities"]),
                torch.zeros_like(ref_p["opacities"]))
    assert ours.params["means3D"].shape[0] == int(keep.sum()) + n_new
    for _ in range(3):
        both_step()
    for k in PARAM_ORDER:
        torch.testing.assert_close(ours.params[k], ref_p[k].detach(), rtol=2e-6, atol=4e-7 * float(ref_p[k].abs().max()))
        torch.testing.assert_close(ours.m[k], ref.state[ref_p[k]]["exp_avg"], rtol=2e-6, atol=1e-7)


def test_live_rows_training_steps_equal_the_dense_path(gpu_device):
    """frg_backward_args::row_live + frg_adam_step_rows: the backward marks the Gaussians with a gradient
    and leaves the rows of the others UNWRITTEN (the gradient buffer is poisoned with NaN before every step to prove it); Adam
    takes an unmarked row as zero without reading it.  Twenty native training steps (raw parameters into the rasterizer,
    fused loss, fused Adam) on a saturating frame (1 M Gaussians on the C3 image: about half of the visible ones without gradient) leave
    parameters and both moments bit-identical to the dense path's."""
    from frosting_amd import scenes
    from frosting_amd.loss import photometric_loss_and_grad
    from frosting_amd.parallel import ViewParallelRasterizer
    dev = gpu_device
    scene, cam, bg = scenes.config_scene("c3", 1, P=1_000_000)
    shapes = {k: tuple(getattr(scene, k).shape) for k in PARAM_ORDER}
    lrs = dict(means3D=1.6e-5, scales=5e-3, rotations=1e-3, opacities=5e-2, shs=2.5e-3)
    cam_d, bg_d = cam.to(dev), bg.to(dev)
    target = None
    results = []
    for live_rows in (False, True):
        opt = FlatAdam(shapes, dict(lrs, shs=lrs["shs"] / 20.0), dev, sh_dc_lr=lrs["shs"])
        opt.params["means3D"].copy_(scene.means3D); opt.params["shs"].copy_(scene.shs)
        opt.params["scales"].copy_(torch.log(scene.scales)); opt.params["rotations"].copy_(scene.rotations * 1.7)
        opt.params["opacities"].copy_(torch.log(scene.opacities / (1 - scene.opacities)))
        raw_scene = scenes.Scene(opt.params["means3D"], opt.params["scales"], opt.params["rotations"], opt.params["opacities"],
                                 opt.params["shs"], scene.sh_degree)
        vpr = ViewParallelRasterizer(raw_scene, dev, raw_params=True, live_rows=live_rows)
        if target is None:
            img, _ = vpr.forward(cam_d, bg_d)
            target = (img + 0.05 * torch.randn(img.shape, generator=torch.Generator().manual_seed(3)).to(dev)).clamp(0, 1)
        fractions = []
        for _ in range(20):
            image, radii = vpr.forward(cam_d, bg_d)
            _, dimg = photometric_loss_and_grad(image, target)
            if live_rows:
                vpr.exchange.flat.fill_(float("nan"))           # whatever is not written must not be read
            vpr.backward(dimg, 0)
            if live_rows:
                vis = radii > 0
                assert not bool(vpr.row_live[~vis].any())
                fractions.append(float(vpr.row_live[vis].float().mean()))
            opt.step(vpr.exchange.flat, row_live=vpr.row_live)
        torch.cuda.synchronize(dev)
        if live_rows:
            # the partially written buffer is refused without its mask; zero_dead_rows makes it the dense form
            with pytest.raises(RuntimeError, match="live_rows"):
                opt.step(vpr.exchange.flat)
            vpr.exchange.flat.fill_(float("nan"))
            vpr.forward(cam_d, bg_d)
            vpr.backward(dimg, 0)
            live_grads = {k: v.clone() for k, v in vpr.zero_dead_rows(0).items()}
            assert all(bool(torch.isfinite(v).all()) for v in live_grads.values()) and bool(torch.isfinite(vpr.dL_dmeans2D).all())
            dense = ViewParallelRasterizer(raw_scene, dev, raw_params=True)
            dense.forward(cam_d, bg_d)
            dense.backward(dimg, 0)
            assert all(torch.equal(live_grads[k], dense.exchange.views[k]) for k in live_grads)
            assert torch.equal(vpr.dL_dmeans2D, dense.dL_dmeans2D)
            assert 0.2 < min(fractions) and max(fractions) < 0.8, fractions         # a saturating frame: many visible Gaussians are never reached
        results.append((opt.flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone()))
    for a, b in zip(*results):
        assert bool(torch.isfinite(b).all())
        assert torch.equal(a, b)

"""One complete native training step of the vanilla 3DGS model at C3 on one MI355X: activations (sigmoid /
exp / normalize of the raw parameters) -> rasterizer forward -> fused photometric loss (value + dL/dimage)
-> rasterizer backward (gradients straight into the flat buffer) -> activations backward, in place on that
buffer -> fused Adam over it (features_dc / features_rest rates on one SH tensor).  The raw parameters live
in FlatAdam's flat buffer.  Prints the time of each part and of the whole step, three times:
  eager   the activations as the reference writes them -- torch.sigmoid / torch.exp / torch.nn.functional.normalize on
          the raw parameters and their autograd backward (frosting_model.py:726-728,770,798; gaussian_model.py:48-57):
          the baseline SURVEY 8(f) rank 3 is measured against;
  launch  one HIP launch for the three activations, one for their backward (frosting_amd/activations.py);
  fused   the activations evaluated inside the per-Gaussian kernels (ViewParallelRasterizer(raw_params=True) ->
          frg_forward_ex / frg_backward_ex on the raw parameters, gradients w.r.t. the raw parameters straight into the
          optimizer's buffer, no activated tensor in memory);
  rows    fused + the backward leaves the rows of Gaussians WITHOUT a gradient unwritten and marks the others
          (frg_backward_args::row_live: six rows in seven at C3), Adam takes an unmarked row as zero without reading it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from frosting_amd import scenes
from frosting_amd.activations import activate, activate_backward_
from frosting_amd.loss import photometric_loss_and_grad
from frosting_amd.optim import FlatAdam
from frosting_amd.parallel import ViewParallelRasterizer, PARAM_ORDER

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else None
scene, cam, bg = scenes.config_scene("c3", 0, P=P)
shapes = {k: tuple(getattr(scene, k).shape) for k in PARAM_ORDER}
lrs = dict(means3D=1.6e-5, scales=5e-3, rotations=1e-3, opacities=5e-2, shs=2.5e-3)   # the reference's rates (OptimizationParams)
opt = FlatAdam(shapes, dict(lrs, shs=lrs["shs"] / 20.0), dev, sh_dc_lr=lrs["shs"])   # features_dc / features_rest rates on one tensor
# raw parameters: log-scale, logit-opacity, unnormalised quaternion (gaussian_model.py:create_from_pcd)
opt.params["means3D"].copy_(scene.means3D); opt.params["shs"].copy_(scene.shs)
opt.params["scales"].copy_(torch.log(scene.scales)); opt.params["rotations"].copy_(scene.rotations * 1.7)
opt.params["opacities"].copy_(torch.log(scene.opacities / (1 - scene.opacities)))
act = tuple(torch.empty_like(opt.params[k]) for k in ("opacities", "scales", "rotations"))
live = scenes.Scene(opt.params["means3D"], act[1], act[2], act[0], opt.params["shs"], scene.sh_degree)


def activations():
    activate(opt.params["opacities"], opt.params["scales"], opt.params["rotations"], out=act)


def activations_backward(gviews):
    activate_backward_(act[0], act[1], opt.params["rotations"], gviews["opacities"], gviews["scales"], gviews["rotations"])


activations()
vpr = ViewParallelRasterizer(live, dev)
cam_d, bg_d = cam.to(dev), bg.to(dev)
img, _ = vpr.forward(cam_d, bg_d)
target = (img + 0.05 * torch.randn_like(img)).clamp(0, 1)
# the same model, raw: the rasterizer reads the optimizer's buffers directly
raw_scene = scenes.Scene(opt.params["means3D"], opt.params["scales"], opt.params["rotations"], opt.params["opacities"],
                         opt.params["shs"], scene.sh_degree)
vpr_raw = ViewParallelRasterizer(raw_scene, dev, raw_params=True)
vpr_rows = ViewParallelRasterizer(raw_scene, dev, raw_params=True, live_rows=True)


# eager: torch leaves aliasing the optimizer's buffers, activated copies kept by autograd
leaves = {k: opt.params[k].detach().requires_grad_(True) for k in ("opacities", "scales", "rotations")}
eager_out = {}


def activations_eager():
    eager_out["o"] = torch.sigmoid(leaves["opacities"])
    eager_out["s"] = torch.exp(leaves["scales"])
    eager_out["q"] = torch.nn.functional.normalize(leaves["rotations"], dim=-1)
    act[0].copy_(eager_out["o"]); act[1].copy_(eager_out["s"]); act[2].copy_(eager_out["q"])      # (what the rasterizer reads)


def activations_backward_eager(gviews):
    go, gs, gq = torch.autograd.grad([eager_out["o"], eager_out["s"], eager_out["q"]],
                                     [leaves["opacities"], leaves["scales"], leaves["rotations"]],
                                     [gviews["opacities"], gviews["scales"], gviews["rotations"]])
    gviews["opacities"].copy_(go); gviews["scales"].copy_(gs); gviews["rotations"].copy_(gq)


def step(raw):
    if raw == "eager":
        activations_eager()
        image, _ = vpr.forward(cam_d, bg_d)
        loss, dimg = photometric_loss_and_grad(image, target)
        g = vpr.backward(dimg, 0)
        activations_backward_eager(g)
        opt.step(vpr.exchange.flat)
        return loss
    if raw == "rows":
        image, _ = vpr_rows.forward(cam_d, bg_d)
        loss, dimg = photometric_loss_and_grad(image, target)
        vpr_rows.backward(dimg, 0)
        opt.step(vpr_rows.exchange.flat, row_live=vpr_rows.row_live)
        return loss
    if raw:
        image, _ = vpr_raw.forward(cam_d, bg_d)
        loss, dimg = photometric_loss_and_grad(image, target)
        vpr_raw.backward(dimg, 0)
        opt.step(vpr_raw.exchange.flat)
        return loss
    activations()
    image, _ = vpr.forward(cam_d, bg_d)
    loss, dimg = photometric_loss_and_grad(image, target)
    g = vpr.backward(dimg, 0)
    activations_backward(g)
    opt.step(vpr.exchange.flat)
    return loss


# every variant starts from the SAME model and optimizer state: 48 Adam steps towards the noisy target change the scene
# (round 3 timed the variants one after the other on the drifting model: its "fused is 0.17 ms slower" was partly that)
snapshot = (opt.flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.steps)
final = {}
for raw in ("eager", False, True, "rows"):
    opt.flat.copy_(snapshot[0]); opt.exp_avg.copy_(snapshot[1]); opt.exp_avg_sq.copy_(snapshot[2]); opt.steps = snapshot[3]
    for _ in range(8):
        step(raw)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
    n, acc = 20, [0.0] * 6
    losses = []
    for _ in range(n):
        if raw == "eager":
            ev[0].record(); activations_eager()
            ev[1].record(); image, _ = vpr.forward(cam_d, bg_d)
            ev[2].record(); loss, dimg = photometric_loss_and_grad(image, target)
            ev[3].record(); g = vpr.backward(dimg, 0)
            ev[4].record(); activations_backward_eager(g)
            ev[5].record(); opt.step(vpr.exchange.flat)
        elif raw == "rows":
            ev[0].record(); ev[1].record(); image, _ = vpr_rows.forward(cam_d, bg_d)
            ev[2].record(); loss, dimg = photometric_loss_and_grad(image, target)
            ev[3].record(); vpr_rows.backward(dimg, 0)
            ev[4].record(); ev[5].record(); opt.step(vpr_rows.exchange.flat, row_live=vpr_rows.row_live)
        elif raw:
            ev[0].record(); ev[1].record(); image, _ = vpr_raw.forward(cam_d, bg_d)
            ev[2].record(); loss, dimg = photometric_loss_and_grad(image, target)
            ev[3].record(); vpr_raw.backward(dimg, 0)
            ev[4].record(); ev[5].record(); opt.step(vpr_raw.exchange.flat)
        else:
            ev[0].record(); activations()
            ev[1].record(); image, _ = vpr.forward(cam_d, bg_d)
            ev[2].record(); loss, dimg = photometric_loss_and_grad(image, target)
            ev[3].record(); g = vpr.backward(dimg, 0)
            ev[4].record(); activations_backward(g)
            ev[5].record(); opt.step(vpr.exchange.flat)
        ev[6].record(); torch.cuda.synchronize()
        for k in range(6):
            acc[k] += ev[k].elapsed_time(ev[k + 1])
        losses.append(float(loss))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step(raw)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    final[raw] = opt.flat.clone()
    what = ("EAGER torch activations + autograd around the rasterizer (the reference's chain)" if raw == "eager" else
            "raw parameters, rows of Gaussians without a gradient neither written nor read (row_live)" if raw == "rows" else
            "raw parameters into the rasterizer (activations inside the per-Gaussian kernels)" if raw else "activation launches around the rasterizer")
    print(f"C3 native training step, P={scene.P}, {what}: activations {acc[0]/n:.3f} ms, forward {acc[1]/n:.3f} ms, loss fwd+bwd "
          f"{acc[2]/n:.3f} ms, backward {acc[3]/n:.3f} ms, activations backward {acc[4]/n:.3f} ms, Adam {acc[5]/n:.3f} ms")
    print(f"    whole step {1e3*t:.3f} ms = {1/t:.0f} steps/s (loss {losses[0]:.5f} -> {losses[-1]:.5f} over {n} steps of the same view)")
print(f"parameters after the 48 steps, rows of Gaussians without gradient unwritten vs the dense fused path: "
      f"{'bit-identical' if torch.equal(final[True], final['rows']) else 'DIFFERENT'}")

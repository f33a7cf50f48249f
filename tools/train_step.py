"""One complete native training step at C3 on one MI355X: rasterizer forward -> fused photometric loss
(value + dL/dimage) -> rasterizer backward (gradients straight into the flat buffer) -> fused Adam over
that buffer.  Parameters live in FlatAdam's flat buffer; the scene tensors are views of it.  Prints the
time of each part and of the whole step.  (Activations / Frosting's parameterisation are not part of it:
SURVEY 8(f) rank 3 is not built.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from frosting_amd import scenes
from frosting_amd.loss import photometric_loss_and_grad
from frosting_amd.optim import FlatAdam
from frosting_amd.parallel import ViewParallelRasterizer, PARAM_ORDER

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else None
scene, cam, bg = scenes.config_scene("c3", 0, P=P)
shapes = {k: tuple(getattr(scene, k).shape) for k in PARAM_ORDER}
lrs = dict(means3D=1.6e-6, scales=5e-5, rotations=1e-5, opacities=5e-4, shs=2.5e-5)   # small: raw (activated) parameters
opt = FlatAdam(shapes, dict(lrs, shs=lrs["shs"] / 20.0), dev, sh_dc_lr=lrs["shs"])   # features_dc / features_rest rates on one tensor
for k in PARAM_ORDER:
    opt.params[k].copy_(getattr(scene, k))
live = scenes.Scene(opt.params["means3D"], opt.params["scales"], opt.params["rotations"], opt.params["opacities"],
                    opt.params["shs"], scene.sh_degree)
vpr = ViewParallelRasterizer(live, dev)
cam_d, bg_d = cam.to(dev), bg.to(dev)
img, _ = vpr.forward(cam_d, bg_d)
target = (img + 0.05 * torch.randn_like(img)).clamp(0, 1)


def step():
    image, _ = vpr.forward(cam_d, bg_d)
    loss, dimg = photometric_loss_and_grad(image, target)
    vpr.backward(dimg, 0)
    opt.step(vpr.exchange.flat)
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
n, acc = 20, [0.0] * 4
t0 = time.perf_counter()
losses = []
for _ in range(n):
    ev[0].record(); image, _ = vpr.forward(cam_d, bg_d)
    ev[1].record(); loss, dimg = photometric_loss_and_grad(image, target)
    ev[2].record(); vpr.backward(dimg, 0)
    ev[3].record(); opt.step(vpr.exchange.flat)
    ev[4].record(); torch.cuda.synchronize()
    for k in range(4):
        acc[k] += ev[k].elapsed_time(ev[k + 1])
    losses.append(float(loss))
t_sync = (time.perf_counter() - t0) / n
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / n
print(f"C3 native training step, P={scene.P}: forward {acc[0]/n:.3f} ms, loss fwd+bwd {acc[1]/n:.3f} ms, "
      f"backward {acc[2]/n:.3f} ms, Adam {acc[3]/n:.3f} ms")
print(f"whole step {1e3*t:.3f} ms = {1/t:.0f} steps/s (loss {losses[0]:.5f} -> {losses[-1]:.5f} over {n} steps of the same view)")

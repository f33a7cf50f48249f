"""Fused Adam at the C3 size (59 floats x 3M Gaussians): time per step and HBM GB/s against
torch.optim.Adam (eager, foreach and fused variants).  Prints a small report."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from frosting_amd.optim import FlatAdam
from frosting_amd.parallel import PARAM_ORDER

dev = torch.device("cuda:0")
P, K = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000, 16
shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
lrs = dict(means3D=1.6e-4, scales=5e-3, rotations=1e-3, opacities=5e-2, shs=2.5e-3)
opt = FlatAdam(shapes, lrs, dev)
opt.flat.normal_()
g = torch.randn(opt.numel, device=dev) * 1e-3


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


t = timeit(lambda: opt.step(g))
print(f"FlatAdam (frg_adam_step): {1e3*t:.3f} ms/step, {28*opt.numel/t/1e9:.0f} GB/s of 28 B/element ({opt.numel/1e6:.0f} M elements)")
opt2 = FlatAdam(shapes, dict(lrs, shs=lrs["shs"] / 20.0), dev, sh_dc_lr=lrs["shs"])
opt2.flat.normal_()
t = timeit(lambda: opt2.step(g))
print(f"FlatAdam with the features_dc / features_rest split on the SH tensor: {1e3*t:.3f} ms/step, {28*opt2.numel/t/1e9:.0f} GB/s")
for name, kw in (("torch Adam single-tensor", dict(foreach=False)), ("torch Adam foreach", dict(foreach=True)),
                 ("torch Adam fused", dict(fused=True))):
    try:
        ps = [torch.nn.Parameter(torch.randn(shapes[k], device=dev)) for k in PARAM_ORDER]
        o = torch.optim.Adam([{"params": [p], "lr": lrs[k]} for p, k in zip(ps, PARAM_ORDER)], lr=0.0, eps=1e-15, **kw)
        for p in ps:
            p.grad = torch.randn_like(p) * 1e-3
        t2 = timeit(o.step)
        print(f"{name}: {1e3*t2:.3f} ms/step")
    except Exception as ex:
        print(f"{name}: not available ({type(ex).__name__})")

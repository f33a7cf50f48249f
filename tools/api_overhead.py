"""Host overhead of the three ways to call the op at C3: persistent-arena C-ABI loop (bench.py's step), the compiled
torch extension through autograd (what the reference's callers do), the ctypes binding through autograd."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from frosting_amd import scenes
from frosting_amd.parallel import ViewParallelRasterizer
from frosting_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer as CtypesRasterizer
from diff_gaussian_rasterization import GaussianRasterizer as ExtRasterizer

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
scene, cam, bg = scenes.config_scene("c3", 0, P=P)
sd, cd, bgd = scene.to(dev), cam.to(dev), bg.to(dev)
vpr = ViewParallelRasterizer(sd, dev)
img, _ = vpr.forward(cd, bgd)
gpix, _ = scenes.l1_target_grad(img.cpu(), 1)
gpix = gpix.to(dev)
settings = GaussianRasterizationSettings(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                         bg=bgd, scale_modifier=1.0, viewmatrix=cd.viewmatrix, projmatrix=cd.projmatrix, sh_degree=3,
                                         campos=cd.campos, prefiltered=False, debug=False)

def run(name, step, n=60):
    for _ in range(20): step()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    walls = []
    evs[0].record()
    for i in range(n):
        t0 = time.perf_counter(); step(); walls.append(time.perf_counter() - t0); evs[i + 1].record()
    torch.cuda.synchronize()
    gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
    print(f"{name:34s} GPU ms/step median {statistics.median(gpu):.4f} mean {statistics.mean(gpu):.4f} max {max(gpu):.3f} | host call ms median {1e3*statistics.median(walls):.4f} mean {1e3*statistics.mean(walls):.4f}", flush=True)

run("C ABI, persistent arenas", lambda: (vpr.forward(cd, bgd), vpr.backward(gpix, 0)))
for name, cls in (("torch extension _C + autograd", ExtRasterizer), ("ctypes binding + autograd", CtypesRasterizer)):
    rast = cls(raster_settings=settings)
    leaves = [t.detach().clone().requires_grad_(True) for t in (sd.means3D, sd.shs, sd.opacities, sd.scales, sd.rotations)]
    m2 = torch.zeros_like(leaves[0], requires_grad=True)
    def step():
        im, _ = rast(means3D=leaves[0], means2D=m2, shs=leaves[1], colors_precomp=None, opacities=leaves[2], scales=leaves[3],
                     rotations=leaves[4], cov3D_precomp=None)
        for t in leaves + [m2]: t.grad = None
        im.backward(gpix)
    run(name, step)
    del rast, leaves, m2

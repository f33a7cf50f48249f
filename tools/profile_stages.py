"""Per-stage GPU times (hipEvents inside the C ABI) for one config.  Usage:
python tools/profile_stages.py [c2|c3] [P] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from frosting_amd import _lib, scenes
from frosting_amd.rasterizer import _C
import helpers as Hh

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
P = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
exact = int(os.environ.get("EXACT", "0"))
dev = torch.device("cuda:0")
scene, cam, bg = scenes.config_scene(cfg, 0, P=P)
_lib.set_option("exact_blend", exact)
_lib.set_option("profile", 1)
_lib.set_option("tight_binning", int(os.environ.get("TIGHT", "0")))
(R, color, radii, geom, binning, img), args = Hh.run_ours_native(scene, cam, bg, dev)
gpix, _ = scenes.l1_target_grad(color.cpu(), 7)
gpix = gpix.to(dev)
acc = {}
wall = []
for it in range(iters + 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(*args)
    bargs = (args[0], args[1], radii, args[2], args[4], args[5], args[6], args[7], args[8], args[9], args[10], args[11],
             gpix, args[14], args[15], args[16], geom, R, binning, img, False)
    _C.rasterize_gaussians_backward(*bargs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if it >= 2:
        wall.append(t1 - t0)
        for k, v in _lib.stage_times().items():
            acc.setdefault(k, []).append(v)
print(f"cfg={cfg} P={scene.P} R={R} V={int((radii>0).sum())} exact={exact}")
tot = 0
for k, v in acc.items():
    print(f"  {k:16s} {np.median(v):8.3f} ms")
    tot += np.median(v)
print(f"  sum of stages    {tot:8.3f} ms ; wall fwd+bwd median {1e3*np.median(wall):.3f} ms")

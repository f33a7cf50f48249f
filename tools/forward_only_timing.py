"""Forward alone at C3 / C2, plain and with frg_forward_args::forward_only (nothing kept for a backward): ms per forward,
alternating, same process.  usage: python tools/forward_only_timing.py [c3|c2 ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from frosting_amd import scenes
from frosting_amd.parallel import ViewParallelRasterizer

dev = torch.device("cuda:0")
for name in (sys.argv[1:] or ["c3", "c2"]):
    scene, cam, bg = scenes.config_scene(name, 0)
    vpr = ViewParallelRasterizer(scene.to(dev), dev)
    cam_d, bg_d = cam.to(dev), bg.to(dev)
    n = 400 if name == "c2" else 60
    res = {False: [], True: []}
    imgs = {}
    for rep in range(3):
        for fo in (False, True):
            for _ in range(10):
                vpr.forward(cam_d, bg_d, forward_only=fo)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n):
                img, _ = vpr.forward(cam_d, bg_d, forward_only=fo)
            torch.cuda.synchronize(dev)
            res[fo].append(1e3 * (time.perf_counter() - t0) / n)
            imgs[fo] = img.clone()
    print(f"{name}: forward alone, plain {min(res[False]):.4f} ms ({', '.join(f'{x:.4f}' for x in res[False])}), "
          f"forward_only {min(res[True]):.4f} ms ({', '.join(f'{x:.4f}' for x in res[True])}); images "
          f"{'bit-identical' if torch.equal(imgs[False], imgs[True]) else 'DIFFERENT'}")

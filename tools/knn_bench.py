"""distCUDA2 at the C3 point count (3M points of the benchmark scene): time of the HIP search."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from frosting_amd import scenes
from frosting_amd.knn import distCUDA2

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
scene, _, _ = scenes.config_scene("c3", 0, P=P)
pts = scene.means3D.to(dev)
for _ in range(2):
    d = distCUDA2(pts)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    d = distCUDA2(pts)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 5
print(f"distCUDA2 on {P} points: {1e3*t:.2f} ms per call; mean 3-NN squared distance {float(d.mean()):.3e}")

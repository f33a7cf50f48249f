"""Which share of the visible Gaussians does a backward have to do mathematics for?  (the sh_dir lead.)

CPU count from the oracle, per scene: V visible; STAGED = Gaussians with an instance inside the walked prefix of its tile's
list (position < the tile's deepest last contributor) -- an upper bound on what the backward blend marks as reached; GRAD =
Gaussians with a non-zero gradient row in the oracle's backward.  The forward's SH pass computes d colour / d direction
(sh_dir, 36 B + ~1000 vector instructions per 64 Gaussians) for all V; a backward that computed it itself would do so for
GRAD of them, reading their 192-byte SH rows.
Analysis tool: uses the oracle, touches nothing of the product.   usage: python tools/reached_fraction.py [P ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from frosting_amd import scenes
from oracle import gs_oracle as G
import helpers as Hh

sizes = [int(a) for a in sys.argv[1:]] or [50_000, 150_000, 400_000, 1_000_000, 3_000_000]
print(f"{'P':>9} {'visible':>9} {'instances':>10} {'walked':>7} {'staged':>9} {'/V':>6} {'grad':>9} {'/V':>6}")
for P in sizes:
    scene, cam, bg = scenes.config_scene("c3", 0, P=P)
    st = G.forward(**Hh.oracle_kwargs(scene, cam, bg))
    W, H = cam.image_width, cam.image_height
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ranges = st["ranges"].astype(np.int64)
    n = st["n_contrib"].astype(np.int64)
    pad = np.zeros((gy * 16, gx * 16), np.int64); pad[:H, :W] = n
    walked = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1)          # per tile
    counts = ranges[:, 1] - ranges[:, 0]
    pos = np.arange(int(counts.sum())) - np.repeat(ranges[:, 0], counts)
    inside = pos < np.repeat(walked, counts)
    staged = np.zeros(P, bool); staged[st["point_list"][inside]] = True
    V = int((st["radii"] > 0).sum())
    gpix, _ = scenes.l1_target_grad(__import__("torch").from_numpy(st["out_color"]), 1)
    g = G.backward(st, gpix.numpy())
    grad = np.zeros(P, bool)
    for k, v in g.items():
        if isinstance(v, np.ndarray) and v.shape[:1] == (P,):
            grad |= (v.reshape(P, -1) != 0).any(1)
    print(f"{P:9d} {V:9d} {int(counts.sum()):10d} {inside.mean():7.3f} {int(staged.sum()):9d} {staged.sum() / V:6.3f} {int(grad.sum()):9d} {grad.sum() / V:6.3f}", flush=True)

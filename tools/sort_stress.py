"""Stress the per-tile sort on the GPU: random tile sizes around every class boundary,
heavy ties, vs numpy lexsort on (tile, depth bits, index)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from frosting_amd import _lib, scenes
from frosting_amd.introspect import State
import helpers as Hh
dev = torch.device("cuda:0")
ok = True
for P, spread, planes in [(500, 0.5, 0), (3000, 0.3, 0), (9000, 0.15, 0), (30000, 0.08, 0), (60000, 0.05, 3), (150000, 0.03, 0)]:
    g = torch.Generator().manual_seed(P)
    cam = scenes.ring_camera(0, 128, 96, 100.0, 100.0)
    means = torch.zeros(P, 3)
    means[:, :2] = spread * torch.randn(P, 2, generator=g)
    means[:, 2] = 0.5 * torch.rand(P, generator=g)
    if planes:
        means[:, 2] = torch.randint(0, planes, (P,), generator=g).float() * 0.1   # massive depth ties
    scene = scenes.Scene(means, torch.full((P, 3), 0.01), torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1),
                         torch.full((P, 1), 0.02), 0.1 * torch.randn(P, 16, 3, generator=g), 3)
    out, _ = Hh.run_ours_native(scene, cam, torch.zeros(3), dev)
    R, color, radii, geom, binning, img = out
    st = State(P, 128, 96, R, geom, binning, img)
    keys = st.sort_keys().cpu().numpy()
    pl = st.point_list.cpu().numpy().astype(np.int64)
    order = np.lexsort((pl, keys))
    good = np.array_equal(order, np.arange(R))
    cnt = st.tile_count.cpu().numpy()
    print(f"P={P} R={R} max tile {cnt.max()} tiles>8192: {(cnt>8192).sum()} sorted(tile,depth,idx): {good}")
    ok &= good
print("ALL OK" if ok else "FAILED")

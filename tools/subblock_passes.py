"""What would packing the forward blend by 4x4 pixel blocks save?  

CPU count over a sample of the C3 frame's tiles, from the oracle's sorted lists and its per-pixel last contributors:
for every 8x8 quadrant the entries its wave walks (up to the last contributor of its last pixel), and per entry which of
its four 4x4 blocks hold a pixel with alpha >= 1/255 (the exact per-block cull; the kernels' closed-form bound keeps a
few more).  Inner-loop passes of one wave (= one 64-lane trip over ~19 vector instructions):
  now        one per (quadrant, entry) pair the quadrant cull keeps;
  rows/round 4 blocks = 4 rows of 16 lanes, every row walks its own compacted list of the staged round of 64 entries,
             the wave's trip count per round = the longest of the four (rows synchronised per round);
  rows/queue the rows run ahead of each other across rounds (per-row queues): trips = the longest row over the whole walk;
  ideal      block pairs / 4.
A block also stops at ITS last contributor (earlier than the quadrant's), which the row forms get for free.
Analysis tool: uses the oracle, touches nothing of the product.   usage: python tools/subblock_passes.py [P] [tiles]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from frosting_amd import scenes
from oracle import gs_oracle as G
import helpers as Hh

P = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 96
scene, cam, bg = scenes.config_scene("c3", 0, P=P)
st = G.forward(**Hh.oracle_kwargs(scene, cam, bg))
W, H = cam.image_width, cam.image_height
gx = (W + 15) // 16
T = st["ranges"].shape[0]
rng = np.random.default_rng(1)
tiles = rng.choice(T, NT, replace=False)
tot = dict(now=0, rows_round=0, rows_queue=0, ideal4=0.0, staged=0, block_pairs=0, quad_pairs=0, hit_px=0)
for t in tiles:
    tx, ty = t % gx, t // gx
    r0, r1 = st["ranges"][t]
    ids = st["point_list"][r0:r1]
    if len(ids) == 0:
        continue
    xy, co = st["means2D"][ids], st["conic_opacity"][ids]
    ncon = st["n_contrib"][ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].astype(np.int64)
    for q in range(4):
        qx, qy = (q & 1) * 8, (q >> 1) * 8
        nq = ncon[qy:qy + 8, qx:qx + 8]
        walk = int(nq.max())                      # entries this quadrant's wave walks
        if walk == 0:
            continue
        px = (tx * 16 + qx + np.arange(8)).astype(np.float32); py = (ty * 16 + qy + np.arange(8)).astype(np.float32)
        dx = xy[:walk, 0, None, None] - px[None, None, :]; dy = xy[:walk, 1, None, None] - py[None, :, None]
        a, b, c, o = (co[:walk, k, None, None] for k in range(4))
        power = -0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy
        hit = (power <= 0) & (np.minimum(0.99, o * np.exp(power)) >= 1.0 / 255.0)          # [walk, 8, 8]
        pos = np.arange(1, walk + 1)[:, None, None]
        live = hit & (pos <= nq[None])            # the pixel had not stopped yet
        qhit = hit.any(axis=(1, 2))               # what the quadrant cull keeps (geometric: does not know about stops)
        tot["now"] += int(qhit.sum()); tot["quad_pairs"] += int(qhit.sum()); tot["staged"] += walk; tot["hit_px"] += int(live.sum())
        # blocks: [walk, 4]; a block's own walk ends at its last contributor
        bh = hit.reshape(walk, 2, 4, 2, 4).any(axis=(2, 4)).reshape(walk, 4)
        bwalk = nq.reshape(2, 4, 2, 4).max(axis=(1, 3)).reshape(4)
        bh = bh & (pos[:, 0, 0, None] <= bwalk[None, :])
        tot["block_pairs"] += int(bh.sum())
        tot["ideal4"] += bh.sum() / 4.0
        tot["rows_queue"] += int(bh.sum(0).max())
        nround = (walk + 63) // 64
        pad = np.zeros((nround * 64, 4), bool); pad[:walk] = bh
        tot["rows_round"] += int(pad.reshape(nround, 64, 4).sum(1).max(1).sum())
print(f"C3 frame, P = {P}, {NT} tiles sampled: per quadrant-wave")
print(f"  entries staged (walked)          {tot['staged']}")
print(f"  inner-loop trips now (8x8 cull)  {tot['now']}   lanes busy {tot['hit_px'] / (64.0 * tot['now']):.3f}")
for k, name in (("rows_round", "4 rows, synchronised per round"), ("rows_queue", "4 rows, queues across rounds "), ("ideal4", "block pairs / 4               ")):
    print(f"  {name}  {tot[k]:.0f}   = {tot[k] / tot['now']:.3f} of now   lanes busy {tot['hit_px'] / (64.0 * tot[k]):.3f}")
print(f"  (block, entry) pairs per kept (quadrant, entry) pair: {tot['block_pairs'] / tot['quad_pairs']:.2f}")

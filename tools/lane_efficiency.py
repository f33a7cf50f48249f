"""How many lanes of a wave pass of the blend kernels do work?  CPU count over a sample of the C3 scene's visible
Gaussians (oracle preprocess, exact alpha >= 1/255 test per pixel): kept (block, Gaussian) pairs and hit pixels for
16x16 tiles, 8x8 quadrants (what the kernels cull by), 8x4 and 4x4 blocks -- the sizing behind DESIGN.md's decision
not to build a sub-block packing of the blends.  Analysis tool: uses the oracle, touches nothing of the product.
usage: python tools/lane_efficiency.py   (a few seconds of the OpenMP oracle + ~20 s of numpy)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frosting_amd import scenes
from oracle import gs_oracle as G
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers as Hh
scene, cam, bg = scenes.config_scene("c3", 0)
kw = Hh.oracle_kwargs(scene, cam, bg)
st = G.forward(stages=("preprocess",), **kw)
rad = st["radii"]; vis = np.nonzero(rad>0)[0]
print("visible", len(vis), "of", scene.P, "R", st["num_rendered"], "mean tiles", st["num_rendered"]/len(vis))
rng = np.random.default_rng(0)
samp = rng.choice(vis, 20000, replace=False)
W,H = cam.image_width, cam.image_height
tot = dict(pix=0, t16=0, q8=0, h84=0, s4=0, tiles_ref=0, tile_hit=0)
hist=[]
for i in samp:
    x,y = st["means2D"][i]; a,b,c,o = st["conic_opacity"][i]; r = rad[i]
    x0=max(0,int((x-r)//16)); x1=min((W+15)//16,int((x+r+15)//16)); y0=max(0,int((y-r)//16)); y1=min((H+15)//16,int((y+r+15)//16))
    if x1<=x0 or y1<=y0: continue
    tot["tiles_ref"] += (x1-x0)*(y1-y0)
    px = np.arange(x0*16, x1*16, dtype=np.float32); py = np.arange(y0*16, y1*16, dtype=np.float32)
    dx = x-px[None,:]; dy = y-py[:,None]
    power = -0.5*(a*dx*dx + c*dy*dy) - b*dx*dy
    alpha = np.minimum(0.99, o*np.exp(power))
    hit = (power<=0)&(alpha>=1/255)
    hit[py>=H,:]=False; hit[:,px>=W]=False
    n = hit.sum(); tot["pix"]+=n; hist.append(n)
    def blocks(bh,bw):
        hh = hit.reshape(hit.shape[0]//bh, bh, hit.shape[1]//bw, bw).any(axis=(1,3))
        return hh.sum()
    tot["t16"]+=blocks(16,16); tot["q8"]+=blocks(8,8); tot["h84"]+=blocks(4,8); tot["s4"]+=blocks(4,4)
print(tot)
print("tiles per G (ref lists):", tot["tiles_ref"]/len(samp), " tiles with any hit:", tot["t16"]/len(samp))
print("quadrant pairs per G:", tot["q8"]/len(samp), " lane eff 8x8:", tot["pix"]/(64*tot["q8"]))
print("8x4 pairs per G:", tot["h84"]/len(samp), " lane eff:", tot["pix"]/(32*tot["h84"]))
print("4x4 pairs per G:", tot["s4"]/len(samp), " lane eff:", tot["pix"]/(16*tot["s4"]))
print("pixels per G mean/median:", np.mean(hist), np.median(hist))
print("wave-passes per G: 8x8:", tot["q8"]/len(samp), " 4x4 packed x4:", tot["s4"]/len(samp)/4, " 8x4 packed x2:", tot["h84"]/len(samp)/2)

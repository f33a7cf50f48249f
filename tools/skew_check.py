import sys,os,time
sys.path.insert(0,'/root/repo' if os.path.isdir('/root/repo') else '.')
import torch
from frosting_amd import scenes,_lib
from frosting_amd.parallel import ViewParallelRasterizer
from frosting_amd.introspect import State
dev=torch.device('cuda:0')
cfg=scenes.CONFIGS['c3']
cam=scenes.ring_camera(0,1600,1056,1334.,1334.).to(dev); bg=torch.zeros(3,device=dev)
sk=scenes.make_skew_scene(3_000_000, cfg['seed']+77).to(dev)
vs=ViewParallelRasterizer(sk,dev)
img,radii=vs.forward(cam,bg)
g,_=scenes.l1_target_grad(img.cpu(),5); g=g.to(dev)
st=State(3_000_000,1600,1056,vs.true_num_rendered,vs.geom.buf,vs.binning.buf,vs.img.buf)
keys=st.sort_keys(); pl=st.point_list.to(torch.int64)
print('sorted', bool((keys[1:]>=keys[:-1]).all()), 'ties ok', bool((pl[1:][keys[1:]==keys[:-1]]>pl[:-1][keys[1:]==keys[:-1]]).all()), 'multiset', bool(torch.equal(torch.bincount(pl,minlength=3_000_000), st.tiles_touched.to(torch.int64))))
for _ in range(5): vs.forward(cam,bg); vs.backward(g,0)
_lib.set_option('profile',1); _lib.stage_times()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): vs.forward(cam,bg); vs.backward(g,0)
torch.cuda.synchronize(); print('skew ms/step', 1e3*(time.perf_counter()-t)/10, {k:round(v,3) for k,v in _lib.stage_times().items() if v>0})

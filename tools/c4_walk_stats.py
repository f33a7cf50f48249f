import sys, torch
sys.path.insert(0, '/root/repo')
from frosting_amd import scenes, mesh as M
from frosting_amd.parallel import ViewParallelRasterizer
from frosting_amd.introspect import State
dev = torch.device('cuda:0')
shell, cam, bg = scenes.config_shell_scene('c4', 0)
scene = shell.scene
vp = ViewParallelRasterizer(scene.to(dev), dev)
cam_d, bg_d = cam.to(dev), bg.to(dev)
ctx = M.RasterizeGLContext()
fm = M.visible_face_mask(shell.verts.to(dev), shell.faces.to(dev), cam_d.projmatrix, cam.image_height, cam.image_width, ctx)
keep = M.occlusion_mask_from_face_mask(shell.cell.to(dev), fm)
img, radii = vp.forward(cam_d, bg_d, keep_mask=keep)
torch.cuda.synchronize()
st = State(scene.P, cam.image_width, cam.image_height, vp.true_num_rendered, vp.geom.buf, vp.binning.buf, vp.img.buf)
n = (st.ranges[:, 1] - st.ranges[:, 0]).float()
# tile_work lives in the image chunk: walked depth per tile = max n_contrib over the tile's pixels
H, W = cam.image_height, cam.image_width
nc = st.n_contrib.view(H // 16, 16, W // 16, 16).amax(dim=(1, 3)).flatten().float()
print('tiles', n.numel(), 'nonempty', int((n > 0).sum()), 'R', int(n.sum()), 'walked', int(nc.sum()))
for q in (0.5, 0.9, 0.99, 0.999, 1.0):
    print(f'list len q{q}: {float(torch.quantile(n, q)):.0f}   walked q{q}: {float(torch.quantile(nc, q)):.0f}')
top = torch.topk(n, 10).indices
print('longest lists -> walked:', [(int(n[i]), int(nc[i])) for i in top])
topw = torch.topk(nc, 10).indices
print('deepest walks -> list:', [(int(nc[i]), int(n[i])) for i in topw])
print('tiles walked > 2000:', int((nc > 2000).sum()), ' > 3000:', int((nc > 3000).sum()), ' > 4000:', int((nc > 4000).sum()))

"""Dump the C4 (shell + occlusion mask) gradients of the current library (FROSTING_LIB honoured) to a file, EXACT
blend -- for comparing two builds bit for bit (tools/c4_grads.py out.pt; python -c 'compare')."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from frosting_amd import _lib, scenes, mesh as M
from frosting_amd.rasterizer import _C

dev = torch.device("cuda:0")
shell, cam, bg = scenes.config_shell_scene("c4", 0, P=2_000_000)
sh = shell.to(dev); sc = sh.scene
H, W = cam.image_height, cam.image_width
fm = M.visible_face_mask(sh.verts, sh.faces, cam.projmatrix.to(dev), H, W)
keep = M.occlusion_mask_from_face_mask(sh.cell, fm)
e = torch.Tensor([])
args = (bg.to(dev), sc.means3D, e, sc.opacities, sc.scales, sc.rotations, 1.0, e, cam.viewmatrix.to(dev),
        cam.projmatrix.to(dev), cam.tanfovx, cam.tanfovy, H, W, sc.shs, 3, cam.campos.to(dev), False, False)
_lib.set_option("exact_blend", 1)
R, color, radii, geom, binning, img = _C.rasterize_gaussians(*args, keep_mask=keep)
gpix, _ = scenes.l1_target_grad(color.cpu(), 41)
gpix = gpix.to(dev)
b = (args[0], args[1], radii, args[2], args[4], args[5], args[6], args[7], args[8], args[9], args[10], args[11],
     gpix, args[14], args[15], args[16], geom, R, binning, img, False)
grads = _C.rasterize_gaussians_backward(*b)
torch.save([g.cpu() for g in grads] + [color.cpu()], sys.argv[1])
print("R", R, "saved", sys.argv[1])

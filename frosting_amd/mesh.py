"""Triangle occlusion raster and the occlusion-culling mask.

Replaces the nvdiffrast seam of the reference: ``frosting_utils/mesh_rasterization.py:6-16``
imports ``nvdiffrast.torch as dr`` inside try/except and ``frosting_utils/nvdiffrast.py:53``
calls ``dr.rasterize(glctx, pos=pos, tri=faces, resolution=[H, W])``.  This module offers
the same two names (``RasterizeGLContext``, ``rasterize``) with the same return convention
on top of the HIP kernel behind ``frg_mesh_rasterize``; ``install_as_nvdiffrast()`` registers
it under ``sys.modules['nvdiffrast.torch']`` so the reference picks it up unchanged.
"""
from __future__ import annotations

import ctypes as C
import sys
import types

import torch

from . import _lib


class RasterizeGLContext:
    """Stand-in for nvdiffrast's OpenGL context object: holds the reusable workspace."""

    def __init__(self, output_db: bool = False, mode: str = "automatic", device=None):
        self.device = device
        self._work = None

    def workspace(self, nbytes, device):
        if self._work is None or self._work.numel() < nbytes or self._work.device != device:
            self._work = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self._work


RasterizeCudaContext = RasterizeGLContext


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    """pos [1,V,4] float32 clip space, tri [F,3] int32, resolution [H,W] ->
    (rast [1,H,W,4] = (u, v, z/w, triangle_id+1), None).  Not differentiable (the
    reference only consumes it under no_grad / for indices)."""
    if pos.dim() != 3 or pos.shape[0] != 1 or pos.shape[2] != 4:
        raise ValueError("pos must be [1, V, 4] (instanced / range mode is not used by the reference)")
    if not pos.is_cuda:
        raise RuntimeError("frosting_amd.mesh.rasterize needs a ROCm device tensor (no CPU path)")
    H, W = int(resolution[0]), int(resolution[1])
    dev = pos.device
    p = pos[0].detach().to(torch.float32).contiguous()
    t = tri.detach().to(device=dev, dtype=torch.int32).contiguous()
    V, F = p.shape[0], t.shape[0]
    L = _lib.lib()
    rast = torch.empty((1, H, W, 4), dtype=torch.float32, device=dev)
    nbytes = int(L.frg_mesh_raster_workspace_bytes(F, W, H))
    ctx = glctx if isinstance(glctx, RasterizeGLContext) else RasterizeGLContext()
    work = ctx.workspace(nbytes, dev)
    with torch.cuda.device(dev):
        rc = L.frg_mesh_rasterize(V, F, C.c_void_p(p.data_ptr()) if V else None, C.c_void_p(t.data_ptr()) if F else None,
                                  W, H, C.c_void_p(rast.data_ptr()), C.c_void_p(work.data_ptr()), work.numel(),
                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc < 0:
        raise RuntimeError(f"frg_mesh_rasterize failed ({rc}): {_lib.last_error()}")
    return rast, None


def fragments(glctx, pos, tri, resolution):
    """The three planes the reference's mesh rasterizer hands on in its nvdiffrast branch
    (frosting_utils/mesh_rasterization.py:146-169, frosting_utils/nvdiffrast.py:53-54), in pytorch3d's `Fragments`
    shapes: bary_coords [1,H,W,1,3] (u, v, 1 - u - v), zbuf [1,H,W,1] (z/w, NDC: the reference leaves it there, :152),
    pix_to_face [1,H,W,1] int32 with -1 for uncovered pixels.  A plain tuple: pytorch3d is not a dependency."""
    H, W = int(resolution[0]), int(resolution[1])
    rast, _ = rasterize(glctx, pos, tri, resolution)
    bary, zbuf, pix_to_face = rast[..., :2], rast[..., 2], rast[..., 3].int() - 1
    bary = torch.cat([bary, 1.0 - bary.sum(dim=-1, keepdim=True)], dim=-1)
    return bary.view(1, H, W, 1, 3), zbuf.view(1, H, W, 1), pix_to_face.view(1, H, W, 1)


def install_as_nvdiffrast():
    """Make ``import nvdiffrast.torch as dr`` resolve to this module."""
    pkg = types.ModuleType("nvdiffrast")
    mod = sys.modules[__name__]
    pkg.torch = mod
    sys.modules["nvdiffrast"] = pkg
    sys.modules["nvdiffrast.torch"] = mod
    return mod


def clip_space_vertices(verts, full_proj_transform):
    """[v,1] @ full_proj_transform, as frosting_utils/nvdiffrast.py:44-50 builds `pos`."""
    ones = torch.ones(verts.shape[0], 1, dtype=verts.dtype, device=verts.device)
    return (torch.cat([verts, ones], dim=1) @ full_proj_transform)[None]


def visible_faces(verts, faces, full_proj_transform, height, width, glctx=None):
    """Indices of the faces seen by at least one pixel (pix_to_face.unique() minus the
    empty marker; frosting_model.py:1534-1539, refine.py:437-441)."""
    rast, _ = rasterize(glctx, clip_space_vertices(verts, full_proj_transform), faces, [height, width])
    ids = rast[..., 3].to(torch.int32).unique() - 1
    return ids[ids >= 0].long()


def visible_face_mask(verts, faces, full_proj_transform, height, width, glctx=None):
    """bool [F]: the face is the nearest surface in at least one pixel.  Same set as visible_faces(), through
    frg_mesh_visible_faces: the z-buffer pass + one byte store per covered pixel, without the attribute resolve, the
    [H,W,4] plane, the sort inside torch.unique or an index_put (the per-frame path of frosting_model.py:1524-1539
    only needs the mask, :1564-1566)."""
    pos = clip_space_vertices(verts, full_proj_transform)[0].detach().to(torch.float32).contiguous()
    if not pos.is_cuda:
        raise RuntimeError("frosting_amd.mesh.visible_face_mask needs ROCm device tensors (no CPU path)")
    dev = pos.device
    t = faces.detach().to(device=dev, dtype=torch.int32).contiguous()
    V, F, H, W = pos.shape[0], t.shape[0], int(height), int(width)
    L = _lib.lib()
    mask = torch.empty(F, dtype=torch.bool, device=dev)
    ctx = glctx if isinstance(glctx, RasterizeGLContext) else RasterizeGLContext()
    work = ctx.workspace(int(L.frg_mesh_raster_workspace_bytes(F, W, H)), dev)
    with torch.cuda.device(dev):
        rc = L.frg_mesh_visible_faces(V, F, C.c_void_p(pos.data_ptr()) if V else None, C.c_void_p(t.data_ptr()) if F else None, W, H,
                                      C.c_void_p(mask.data_ptr()) if F else None, C.c_void_p(work.data_ptr()), work.numel(),
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc < 0:
        raise RuntimeError(f"frg_mesh_visible_faces failed ({rc}): {_lib.last_error()}")
    return mask


def occlusion_keep_mask(verts, faces, full_proj_transform, height, width, point_cell_indices, n_background=0, glctx=None,
                        return_face_mask=False):
    """The keep mask of one frame in ONE native call (frg_mesh_occlusion_mask): what
    occlusion_mask_from_face_mask(point_cell_indices, visible_face_mask(verts, faces, full_proj_transform, ...), n_background)
    composes from torch.ones / cat / matmul, the z-buffer pass and a torch index (frosting_model.py:1524-1539, 1564-1586) --
    five back-to-back launches instead of nine with host work between them.  bool [len(point_cell_indices) + n_background]."""
    v = verts.detach().to(torch.float32).contiguous()
    if not v.is_cuda:
        raise RuntimeError("frosting_amd.mesh.occlusion_keep_mask needs ROCm device tensors (no CPU path)")
    dev = v.device
    t = faces.detach().to(device=dev, dtype=torch.int32).contiguous()
    m = full_proj_transform.detach().to(device=dev, dtype=torch.float32).contiguous()
    cells = point_cell_indices.detach().to(device=dev, dtype=torch.int64).contiguous()
    V, F, H, W, n_shell = v.shape[0], t.shape[0], int(height), int(width), cells.shape[0]
    L = _lib.lib()
    face_mask = torch.empty(F, dtype=torch.bool, device=dev)
    keep = torch.empty(n_shell + int(n_background), dtype=torch.bool, device=dev)
    ctx = glctx if isinstance(glctx, RasterizeGLContext) else RasterizeGLContext()
    work = ctx.workspace(int(L.frg_mesh_occlusion_workspace_bytes(V, F, W, H)), dev)
    ptr = lambda x: C.c_void_p(x.data_ptr()) if x.numel() else None
    with torch.cuda.device(dev):
        rc = L.frg_mesh_occlusion_mask(V, F, ptr(v), ptr(m), ptr(t), W, H, n_shell, ptr(cells), int(n_background), ptr(keep),
                                       ptr(face_mask), C.c_void_p(work.data_ptr()), work.numel(),
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc < 0:
        raise RuntimeError(f"frg_mesh_occlusion_mask failed ({rc}): {_lib.last_error()}")
    return (keep, face_mask) if return_face_mask else keep


def occlusion_mask_from_face_mask(point_cell_indices, face_mask, n_background=0):
    """occlusion_mask() for a precomputed boolean face mask."""
    keep = face_mask[point_cell_indices]
    if n_background:
        keep = torch.cat([keep, torch.ones(n_background, dtype=torch.bool, device=keep.device)])
    return keep


def occlusion_mask(point_cell_indices, face_idx_to_render, n_faces, n_background=0):
    """Per-Gaussian keep mask of the Frosting occlusion culling (frosting_model.py:1564-1586):
    a shell Gaussian survives iff the base face of its cell is visible; background Gaussians
    are always kept."""
    face_mask = torch.zeros(n_faces, dtype=torch.bool, device=point_cell_indices.device)
    face_mask[face_idx_to_render] = True
    keep = face_mask[point_cell_indices]
    if n_background:
        keep = torch.cat([keep, torch.ones(n_background, dtype=torch.bool, device=keep.device)])
    return keep


def depth_keep_mask(means3D, viewmatrix, projmatrix, depth, tolerance):
    """Per-Gaussian keep mask of Frosting's OTHER culling variant -- against a depth map instead of the set of visible faces
    (frosting_scene/frosting_model.py:70-135 get_points_depth_in_depthmaps, :1547-1562): a Gaussian survives iff it projects
    inside the image AND (its view-space depth is in front of the depth map's value at its projection + tolerance, OR the
    map holds no depth there: map_z <= 0).  The map is sampled as the reference samples it -- torch.nn.functional.grid_sample,
    bilinear, zeros outside, align_corners=False -- at the Gaussian's centre.

    means3D [P,3]; viewmatrix, projmatrix: the rasterizer's row-vector matrices (world -> view, world -> clip; the camera the
    depth map was rendered with -- e.g. fragments(...).zbuf of this module's triangle raster, 0 where no triangle covers the
    pixel); depth [H,W]; tolerance: absolute, in view-space units (the reference passes filtering_tolerance x the scene's
    spatial extent).  Returns bool [P].  Plain torch on whatever device the tensors live on: this variant is host-side glue
    in the reference too (non-default: refine.py:45).  The reference reaches the same quantities through pytorch3d cameras
    (their NDC has +x left / +y up and is rescaled by -min(H, W) / W | H before grid_sample: frosting_model.py:107-111), which
    this image does not have: the function follows the reference's formulas, its parity against the reference is UNPINNED."""
    m = means3D.detach().to(torch.float32)
    ones = torch.ones((m.shape[0], 1), dtype=m.dtype, device=m.device)
    hom = torch.cat([m, ones], 1)
    real_z = (hom @ viewmatrix.to(m))[:, 2]
    clip = hom @ projmatrix.to(m)
    ndc = clip[:, :2] / (clip[:, 3:4] + 1e-7)                    # x right, y down, [-1, 1] across the image: grid_sample's convention
    inside = ~((ndc.min(-1)[0] < -1) | (ndc.max(-1)[0] > 1))
    map_z = torch.nn.functional.grid_sample(depth.to(m)[None, None], ndc.view(1, -1, 1, 2), mode="bilinear", padding_mode="zeros",
                                            align_corners=False)[0, 0, :, 0]
    return ((real_z < map_z + float(tolerance)) | (map_z <= 0.0)) & inside

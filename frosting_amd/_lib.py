"""ctypes binding of libfrosting_rasterizer.so (include/frosting_rasterizer.h).

There is no fallback: if the HIP library is missing or cannot be loaded the
import-time / first-use error is fatal by design (a silent CPU or eager path
would void every parity and performance claim).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# FROSTING_LIB: another build of the same library (A/B timing of kernel variants in one GPU session; tools/ab.py)
LIB_PATH = os.environ.get("FROSTING_LIB") or os.path.join(_PKG, "lib", "libfrosting_rasterizer.so")
CSRC = os.path.join(_PKG, "csrc")

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
ECAPACITY = -5

_lib = None


class ForwardArgs(C.Structure):
    """frg_forward_args (include/frosting_rasterizer.h)."""
    _fields_ = [("struct_size", C.c_size_t),
                ("geometry_alloc", ALLOC_FN), ("binning_alloc", ALLOC_FN), ("image_alloc", ALLOC_FN),
                ("user", C.c_void_p),
                ("P", C.c_int), ("D", C.c_int), ("M", C.c_int),
                ("background", C.c_void_p),
                ("width", C.c_int), ("height", C.c_int),
                ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
                ("opacities", C.c_void_p), ("scales", C.c_void_p),
                ("scale_modifier", C.c_float),
                ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p),
                ("projmatrix", C.c_void_p), ("cam_pos", C.c_void_p),
                ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
                ("prefiltered", C.c_int),
                ("out_color", C.c_void_p), ("radii", C.c_void_p),
                ("debug", C.c_int),
                ("hip_stream", C.c_void_p),
                ("instance_capacity", C.c_int),
                ("keep_mask", C.c_void_p),
                # raw-parameter mode (activations and the shell gather inside the per-Gaussian kernels)
                ("raw_opacities", C.c_void_p), ("raw_scales", C.c_void_p), ("raw_rotations", C.c_void_p),
                ("shell_logits", C.c_void_p), ("shell_cell_verts", C.c_void_p), ("shell_cells", C.c_void_p),
                # per-call modes: 0 = the process-wide frg_set_option value, k + 1 = value k for this call
                ("exact_blend", C.c_int), ("tight_binning", C.c_int), ("async_sh", C.c_int),
                ("shell_bary_mode", C.c_int),
                # no backward will follow: nothing is kept for one
                ("forward_only", C.c_int)]


def mode_fields(modes) -> dict:
    """{'exact_blend': 0|1, 'tight_binning': 0|1, 'async_sh': 0..3} (any subset, or None) -> the per-call mode fields
    of frg_forward_args (0 = process default, k + 1 = value k)."""
    out = {"exact_blend": 0, "tight_binning": 0, "async_sh": 0}
    forward_only = 0
    for k, v in (modes or {}).items():
        if k == "forward_only":          # a plain 0 | 1 field, no process-wide form
            forward_only = int(bool(v))
            continue
        if k not in out:
            raise KeyError(f"unknown per-call mode '{k}'")
        out[k] = int(v) + 1
    out["forward_only"] = forward_only
    return out


class BackwardArgs(C.Structure):
    """frg_backward_args (include/frosting_rasterizer.h)."""
    _fields_ = [("struct_size", C.c_size_t),
                ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("R", C.c_int),
                ("background", C.c_void_p),
                ("width", C.c_int), ("height", C.c_int),
                ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p), ("scales", C.c_void_p),
                ("scale_modifier", C.c_float),
                ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p),
                ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
                ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
                ("radii", C.c_void_p),
                ("geom_buffer", C.c_void_p), ("binning_buffer", C.c_void_p), ("image_buffer", C.c_void_p),
                ("dL_dpix", C.c_void_p),
                ("dL_dmean2D", C.c_void_p), ("dL_dconic", C.c_void_p), ("dL_dopacity", C.c_void_p), ("dL_dcolor", C.c_void_p),
                ("dL_dmean3D", C.c_void_p), ("dL_dcov3D", C.c_void_p), ("dL_dsh", C.c_void_p), ("dL_dscale", C.c_void_p),
                ("dL_drot", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("debug", C.c_int),
                ("hip_stream", C.c_void_p),
                ("raw_opacities", C.c_void_p), ("raw_scales", C.c_void_p), ("raw_rotations", C.c_void_p),
                ("shell_logits", C.c_void_p), ("shell_cell_verts", C.c_void_p), ("shell_cells", C.c_void_p),
                ("dL_dshell_logits", C.c_void_p), ("dL_dshell_cell_verts", C.c_void_p),
                ("exact_blend", C.c_int), ("shell_bary_mode", C.c_int),
                # the backward in two calls: 1 = blend + slot sums (dL_dcolor complete), 2 = the rest; 0 = one call
                ("phase", C.c_int),
                # optional [P] bytes: 1 = the Gaussian has a gradient; the rows of the others are then not written
                ("row_live", C.c_void_p),
                # phase 1 in pieces: Gaussians [range_first, + range_count) (range_count 0: the whole phase)
                ("range_first", C.c_int), ("range_count", C.c_int)]


class CombineArgs(C.Structure):
    """frg_combine_args (include/frosting_rasterizer.h)."""
    _fields_ = [("struct_size", C.c_size_t),
                ("P", C.c_int), ("first", C.c_int), ("count", C.c_int), ("n_views", C.c_int),
                ("packets", C.c_void_p),
                ("packet_stride_bytes", C.c_size_t),
                ("capacity_rows", C.c_longlong),
                ("M", C.c_int),
                ("means3D", C.c_void_p), ("shs", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("opacities", C.c_void_p),
                ("raw_opacities", C.c_void_p), ("raw_scales", C.c_void_p), ("raw_rotations", C.c_void_p),
                ("dL_dmean3D", C.c_void_p), ("dL_dscale", C.c_void_p), ("dL_drot", C.c_void_p), ("dL_dopacity", C.c_void_p),
                ("dL_dsh", C.c_void_p),
                ("status", C.c_void_p),
                ("status_seq", C.c_uint),
                ("row_live", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("hip_stream", C.c_void_p)]


def build(verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles on CPU-only hosts)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=out)
    return LIB_PATH


class RasterizerLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RasterizerLibraryError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C {CSRC}).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.frg_version.restype = i
    L.frg_last_error.restype = C.c_char_p
    L.frg_set_option.argtypes = [C.c_char_p, i]
    L.frg_get_option.argtypes = [C.c_char_p]
    L.frg_geometry_bytes.restype = sz
    L.frg_geometry_bytes.argtypes = [i]
    L.frg_image_bytes.restype = sz
    L.frg_image_bytes.argtypes = [i, i]
    L.frg_binning_bytes.restype = sz
    L.frg_binning_bytes.argtypes = [i, i]
    L.frg_backward_workspace_bytes.restype = sz
    L.frg_backward_workspace_bytes.argtypes = [i, i]
    L.frg_geometry_layout.argtypes = [i, vp]
    if hasattr(L, "frg_geometry_layout_n"):        # (absent from a version-1 library loaded through FROSTING_LIB for an A/B)
        L.frg_geometry_layout_n.restype = i
        L.frg_geometry_layout_n.argtypes = [i, vp, i]
    L.frg_image_layout.argtypes = [i, i, vp]
    L.frg_binning_layout.argtypes = [i, i, vp]
    L.frg_mark_visible.restype = i
    L.frg_mark_visible.argtypes = [i, vp, vp, vp, vp, vp]
    L.frg_forward.restype = i
    L.frg_forward.argtypes = [ALLOC_FN, ALLOC_FN, ALLOC_FN, vp,
                              i, i, i, vp, i, i,
                              vp, vp, vp, vp,
                              vp, f, vp, vp,
                              vp, vp, vp,
                              f, f, i,
                              vp, vp, i, vp]
    L.frg_forward_deferred.restype = i
    L.frg_forward_deferred.argtypes = list(L.frg_forward.argtypes)   # `debug` slot carries instance_capacity
    L.frg_knn_workspace_bytes.restype = sz
    L.frg_knn_workspace_bytes.argtypes = [i]
    L.frg_knn_mean_dist2.restype = i
    L.frg_knn_mean_dist2.argtypes = [i, vp, vp, vp, sz, vp]
    L.frg_shell_points.restype = i
    L.frg_shell_points.argtypes = [i, vp, vp, vp, vp, vp]
    L.frg_shell_points_backward.restype = i
    L.frg_shell_points_backward.argtypes = [i, vp, vp, vp, vp, vp, vp]
    L.frg_activate.restype = i
    L.frg_activate.argtypes = [i, vp, vp, vp, vp, vp, vp, vp]
    L.frg_activate_backward.restype = i
    L.frg_activate_backward.argtypes = [i, vp, vp, vp, vp, vp, vp, vp]
    L.frg_photometric_workspace_bytes.restype = sz
    L.frg_photometric_workspace_bytes.argtypes = [i, i, i]
    L.frg_photometric_loss.restype = i
    L.frg_photometric_loss.argtypes = [i, i, i, vp, vp, C.POINTER(C.c_float), f, vp, vp, vp, sz, vp]
    L.frg_adam_step.restype = i
    L.frg_adam_step.argtypes = [C.c_longlong, vp, vp, vp, vp, C.POINTER(C.c_longlong), C.POINTER(C.c_float),
                                C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), i,
                                C.c_double, C.c_double, C.c_double, i, f, vp]
    L.frg_forward_ex.restype = i
    L.frg_forward_ex.argtypes = [C.POINTER(ForwardArgs)]
    L.frg_backward_ex.restype = i
    L.frg_backward_ex.argtypes = [C.POINTER(BackwardArgs)]
    L.frg_forward_finish.restype = i
    L.frg_forward_finish.argtypes = [vp, i, C.POINTER(C.c_int)]
    L.frg_backward.restype = i
    L.frg_backward.argtypes = [i, i, i, i, vp, i, i,
                               vp, vp, vp,
                               vp, f, vp, vp,
                               vp, vp, vp,
                               f, f, vp,
                               vp, vp, vp, vp,
                               vp, vp, vp, vp,
                               vp, vp, vp, vp, vp,
                               vp, sz, i, vp]
    L.frg_stage_times.restype = i
    L.frg_stage_times.argtypes = [vp, i]
    L.frg_mesh_raster_workspace_bytes.restype = sz
    L.frg_mesh_raster_workspace_bytes.argtypes = [i, i, i]
    L.frg_mesh_rasterize.restype = i
    L.frg_mesh_rasterize.argtypes = [i, i, vp, vp, i, i, vp, vp, sz, vp]
    L.frg_mesh_occlusion_workspace_bytes.restype = sz
    L.frg_mesh_occlusion_workspace_bytes.argtypes = [i, i, i, i]
    L.frg_mesh_occlusion_mask.restype = i
    L.frg_mesh_occlusion_mask.argtypes = [i, i, vp, vp, vp, i, i, i, vp, i, vp, vp, vp, sz, vp]
    if hasattr(L, "frg_mesh_visible_faces"):
        L.frg_mesh_visible_faces.restype = i
        L.frg_mesh_visible_faces.argtypes = [i, i, vp, vp, i, i, vp, vp, sz, vp]
    L.frg_sh_color_grad.restype = i
    L.frg_sh_color_grad.argtypes = [i, vp, vp, vp, vp, vp]
    L.frg_sh_grad_from_views.restype = i
    L.frg_sh_grad_from_views.argtypes = [i, i, i, i, vp, vp, C.c_longlong, vp, C.c_longlong, vp, vp]
    L.frg_adam_step_shard.restype = i
    L.frg_adam_step_shard.argtypes = [C.c_longlong, C.c_longlong, vp, vp, vp, vp, C.POINTER(C.c_longlong), C.POINTER(C.c_float),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), i,
                                      C.c_double, C.c_double, C.c_double, i, f, vp]
    L.frg_adam_step_rows.restype = i
    L.frg_adam_step_rows.argtypes = [C.c_longlong, vp, vp, vp, vp, C.POINTER(C.c_longlong), C.POINTER(C.c_float),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), i,
                                     C.c_double, C.c_double, C.c_double, i, f, vp, i, C.POINTER(C.c_int), vp]
    L.frg_pack_grad_rows.restype = i
    L.frg_pack_grad_rows.argtypes = [i, vp, vp, vp, vp, vp, vp, C.c_longlong, vp, vp]
    L.frg_scatter_grad_rows.restype = i
    L.frg_scatter_grad_rows.argtypes = [C.c_longlong, i, vp, vp, vp, vp, vp, vp, vp]
    L.frg_sum_packet_bytes.restype = sz
    L.frg_sum_packet_bytes.argtypes = [i, C.c_longlong]
    L.frg_pack_sum_rows.restype = i
    L.frg_pack_sum_rows.argtypes = [i, i, i, i, vp, sz, vp, vp, vp, vp, f, f, i, i, f, i, vp, sz, C.c_longlong, vp]
    L.frg_combine_workspace_bytes.restype = sz
    L.frg_combine_workspace_bytes.argtypes = [i, C.c_longlong]
    L.frg_backward_combine.restype = i
    L.frg_backward_combine.argtypes = [C.POINTER(CombineArgs)]
    _lib = L
    return L


def build_fingerprint() -> dict:
    """What identifies the kernels a measurement was taken with: sha256 over the kernel sources (csrc/*.hip, *.h, the
    Makefile and the public header, sorted by name -- the same wherever the tree is checked out) and over the loaded
    library file.  tools/collect_traffic.py stores it beside the PMC figures; bench.py only quotes figures whose
    source hash is that of the running tree."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) or f == "Makefile")
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(os.path.dirname(_PKG), "include", "frosting_rasterizer.h"), "rb").read())
    lib_sha = hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest() if os.path.exists(LIB_PATH) else None
    return {"kernel_sources_sha256": h.hexdigest(), "library_sha256": lib_sha}


def last_error() -> str:
    return lib().frg_last_error().decode("utf-8", "replace")


def set_option(name: str, value: int) -> int:
    return lib().frg_set_option(name.encode(), int(value))


def get_option(name: str) -> int:
    return lib().frg_get_option(name.encode())


STAGE_NAMES = ["preprocess", "scan", "scatter", "sort", "blend_fwd", "blend_bwd", "preprocess_bwd", "sh_color"]


def stage_times() -> dict:
    """Milliseconds per stage of the last forward/backward (needs set_option('profile', 1))."""
    buf = (C.c_float * 16)()
    n = lib().frg_stage_times(buf, 16)
    return {STAGE_NAMES[k]: float(buf[k]) for k in range(min(n, len(STAGE_NAMES)))}


EXPORTED_SYMBOLS = [
    "frg_version", "frg_last_error", "frg_mark_visible", "frg_forward", "frg_backward_workspace_bytes",
    "frg_backward", "frg_set_option", "frg_get_option", "frg_stage_times", "frg_geometry_bytes", "frg_image_bytes",
    "frg_binning_bytes", "frg_geometry_layout", "frg_geometry_layout_n", "frg_image_layout", "frg_binning_layout",
    "frg_mesh_raster_workspace_bytes", "frg_mesh_rasterize", "frg_mesh_visible_faces", "frg_mesh_occlusion_workspace_bytes", "frg_mesh_occlusion_mask", "frg_sh_color_grad", "frg_sh_grad_from_views",
    "frg_pack_grad_rows", "frg_scatter_grad_rows", "frg_adam_step_rows", "frg_adam_step_shard",
    "frg_sum_packet_bytes", "frg_pack_sum_rows", "frg_combine_workspace_bytes", "frg_backward_combine",
    "frg_forward_deferred", "frg_forward_finish", "frg_forward_ex", "frg_backward_ex", "frg_adam_step",
    "frg_photometric_workspace_bytes", "frg_photometric_loss", "frg_activate", "frg_activate_backward",
    "frg_knn_workspace_bytes", "frg_knn_mean_dist2", "frg_shell_points", "frg_shell_points_backward",
]

"""TEST INFRASTRUCTURE: runs the *reference's own* rasterizer (oracle/_ref, built by
oracle/build_ref.sh from /root/reference with hipcc) on the GPU through ctypes.

Used only by tests/ and tools/ (parity checks, golden-fixture generation, the
"reference on MI355X" timing beside ours).  Never imported by the product path.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
_libs = {}


def lib_path(variant: str = "exact") -> str:
    return os.path.join(_HERE, "_ref", f"libref_rasterizer_{variant}.so")


def available(variant: str = "exact") -> bool:
    return os.path.exists(lib_path(variant))


def lib(variant: str = "exact"):
    if variant not in _libs:
        L = C.CDLL(lib_path(variant))
        L.ref_forward.restype = C.c_int
        _libs[variant] = L
    return _libs[variant]


def _p(t):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


class RefState:
    """Decoded view of the reference's three opaque chunks (rasterizer_impl.h:21-73)."""

    def __init__(self, L, P, W, H, R, geom, binning, img):
        self.P, self.W, self.H, self.R = P, W, H, R
        self.geom, self.binning, self.img = geom, binning, img
        off = (C.c_longlong * 16)()
        N = W * H

        def view(buf, o, dtype, n):
            esz = torch.empty(0, dtype=dtype).element_size()
            base = buf.data_ptr()
            start = int(o)
            return buf[start:start + n * esz].view(dtype)

        L.ref_geom_offsets(C.c_void_p(geom.data_ptr()), C.c_size_t(P), off)
        self.depths = view(geom, off[0], torch.float32, P)
        self.clamped = view(geom, off[1], torch.uint8, 3 * P).view(P, 3)
        self.means2D = view(geom, off[3], torch.float32, 2 * P).view(P, 2)
        self.cov3D = view(geom, off[4], torch.float32, 6 * P).view(P, 6)
        self.conic_opacity = view(geom, off[5], torch.float32, 4 * P).view(P, 4)
        self.rgb = view(geom, off[6], torch.float32, 3 * P).view(P, 3)
        self.tiles_touched = view(geom, off[7], torch.int32, P)
        self.point_offsets = view(geom, off[8], torch.int32, P)
        L.ref_image_offsets(C.c_void_p(img.data_ptr()), C.c_size_t(N), off)
        self.final_T = view(img, off[0], torch.float32, N).view(H, W)
        self.n_contrib = view(img, off[1], torch.int32, N).view(H, W)
        T = ((W + 15) // 16) * ((H + 15) // 16)
        self.ranges = view(img, off[2], torch.int32, 2 * T).view(T, 2)
        if R > 0:
            L.ref_binning_offsets(C.c_void_p(binning.data_ptr()), C.c_size_t(R), off)
            self.point_list = view(binning, off[0], torch.int32, R)
            self.point_list_keys = view(binning, off[2], torch.int64, R)
        else:
            self.point_list = torch.empty(0, dtype=torch.int32, device=geom.device)
            self.point_list_keys = torch.empty(0, dtype=torch.int64, device=geom.device)


def forward(means3D, opacities, viewmatrix, projmatrix, campos, bg, width, height, tanfovx, tanfovy,
            shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=0,
            scale_modifier=1.0, variant="exact", debug=False):
    """All tensors float32 CUDA.  Returns (num_rendered, out_color[3,H,W], radii[P], RefState)."""
    L = lib(variant)
    dev = means3D.device
    P, H, W = means3D.shape[0], int(height), int(width)
    out_color = torch.zeros(3, H, W, dtype=torch.float32, device=dev)
    radii = torch.zeros(P, dtype=torch.int32, device=dev)
    bufs = {}

    def mk(name):
        def cb(_u, n):
            bufs[name] = torch.empty(max(int(n), 1) + 256, dtype=torch.uint8, device=dev)
            return bufs[name].data_ptr()
        return ALLOC_FN(cb)

    cbs = [mk("geom"), mk("binning"), mk("img")]
    M = 0 if shs is None else shs.shape[1]
    cont = lambda t: None if t is None else t.contiguous()
    t = [cont(x) for x in (bg, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix,
                           projmatrix, campos)]
    torch.cuda.synchronize(dev)
    R = L.ref_forward(cbs[0], cbs[1], cbs[2], None, C.c_int(P), C.c_int(int(sh_degree)), C.c_int(M), _p(t[0]),
                      C.c_int(W), C.c_int(H), _p(t[1]), _p(t[2]), _p(t[3]), _p(t[4]), _p(t[5]),
                      C.c_float(scale_modifier), _p(t[6]), _p(t[7]), _p(t[8]), _p(t[9]), _p(t[10]),
                      C.c_float(tanfovx), C.c_float(tanfovy), C.c_int(0), _p(out_color), _p(radii), C.c_int(int(debug)))
    torch.cuda.synchronize(dev)  # the reference launches on the legacy default stream
    st = RefState(L, P, W, H, R, bufs["geom"], bufs.get("binning", torch.empty(1, dtype=torch.uint8, device=dev)),
                  bufs["img"])
    st.inputs = dict(bg=t[0], means3D=t[1], shs=t[2], colors_precomp=t[3], scales=t[5], rotations=t[6],
                     cov3D_precomp=t[7], viewmatrix=t[8], projmatrix=t[9], campos=t[10], tanfovx=float(tanfovx),
                     tanfovy=float(tanfovy), sh_degree=int(sh_degree), scale_modifier=float(scale_modifier), M=M,
                     variant=variant)
    st.radii = radii
    return R, out_color, radii, st


def backward(st: RefState, dL_dout_color, debug=False):
    """Reference backward (atomics => summation order varies run to run)."""
    i = st.inputs
    L = lib(i["variant"])
    dev = st.geom.device
    P, W, H, R, M = st.P, st.W, st.H, st.R, i["M"]
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
    g = dict(dL_dmeans2D=z(P, 3), dL_dconic=z(P, 4), dL_dopacity=z(P, 1), dL_dcolors=z(P, 3), dL_dmeans3D=z(P, 3),
             dL_dcov3D=z(P, 6), dL_dsh=z(P, M, 3), dL_dscales=z(P, 3), dL_drotations=z(P, 4))
    dpix = dL_dout_color.contiguous()
    torch.cuda.synchronize(dev)
    L.ref_backward(C.c_int(P), C.c_int(i["sh_degree"]), C.c_int(M), C.c_int(R), _p(i["bg"]), C.c_int(W), C.c_int(H),
                   _p(i["means3D"]), _p(i["shs"]), _p(i["colors_precomp"]), _p(i["scales"]),
                   C.c_float(i["scale_modifier"]), _p(i["rotations"]), _p(i["cov3D_precomp"]), _p(i["viewmatrix"]),
                   _p(i["projmatrix"]), _p(i["campos"]), C.c_float(i["tanfovx"]), C.c_float(i["tanfovy"]),
                   _p(st.radii), _p(st.geom), _p(st.binning), _p(st.img), _p(dpix), _p(g["dL_dmeans2D"]),
                   _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]), _p(g["dL_dmeans3D"]),
                   _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]), C.c_int(int(debug)))
    torch.cuda.synchronize(dev)
    return g

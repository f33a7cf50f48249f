"""TEST INFRASTRUCTURE: ctypes/numpy front end of the C restatement (gs_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It is the checker, never the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgs_oracle.so")
_lib = None


_LIB64_PATH = os.path.join(_HERE, "_build", "libgs_oracle_f64.so")
_lib64 = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gs_oracle.c")
    for path in (_LIB_PATH, _LIB64_PATH):
        if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, os.path.relpath(path, _HERE)], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib64():
    """gs_oracle.c with every float a double (oracle/Makefile, F64): only its two backward functions are used."""
    global _lib64
    if _lib64 is None:
        if not os.path.exists(_LIB64_PATH):
            build()
        _lib64 = C.CDLL(_LIB64_PATH)
    return _lib64


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gso_preprocess.restype = C.c_int
        _lib.gso_num_threads.restype = C.c_int
    return _lib


def _f(a):
    if a is None:
        return None
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def num_threads() -> int:
    return int(lib().gso_num_threads())


def set_num_threads(n: int):
    lib().gso_set_num_threads(int(n))


def mark_visible(means3D, viewmatrix, projmatrix):
    m, v, pm = _f(means3D), _f(viewmatrix), _f(projmatrix)
    P = m.shape[0]
    out = np.zeros(P, dtype=np.uint8)
    lib().gso_mark_visible(C.c_int(P), _p(m), _p(v), _p(pm), _p(out))
    return out.astype(bool)


def forward(means3D, opacities, viewmatrix, projmatrix, campos, bg, width, height, tanfovx, tanfovy,
            shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            sh_degree=0, scale_modifier=1.0, stages=("preprocess", "sort", "render")):
    """Run the restated forward.  Returns a dict with every artefact the
    reference materialises (names follow rasterizer_impl.h:21-73)."""
    L = lib()
    m = _f(means3D)
    P = m.shape[0]
    H, W = int(height), int(width)
    sh, cp, sc, rot, c3p = _f(shs), _f(colors_precomp), _f(scales), _f(rotations), _f(cov3D_precomp)
    op, vm, pm, cam, bgc = _f(opacities), _f(viewmatrix), _f(projmatrix), _f(campos).reshape(-1), _f(bg)
    M = 0 if sh is None else sh.shape[1]
    st = dict(P=P, W=W, H=H, M=M, D=int(sh_degree))
    st["radii"] = np.zeros(P, np.int32)
    st["means2D"] = np.zeros((P, 2), np.float32)
    st["depths"] = np.zeros(P, np.float32)
    st["cov3D"] = np.zeros((P, 6), np.float32)
    st["rgb"] = np.zeros((P, 3), np.float32)
    st["conic_opacity"] = np.zeros((P, 4), np.float32)
    st["clamped"] = np.zeros((P, 3), np.uint8)
    st["tiles_touched"] = np.zeros(P, np.uint32)
    st["point_offsets"] = np.zeros(P, np.uint32)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    st["grid"] = (gx, gy)
    if P == 0:
        st["num_rendered"] = 0
        st["out_color"] = np.zeros((3, H, W), np.float32)
        return st
    R = L.gso_preprocess(C.c_int(P), C.c_int(int(sh_degree)), C.c_int(M), _p(m), _p(sc), C.c_float(scale_modifier),
                         _p(rot), _p(op), _p(sh), _p(c3p), _p(cp), _p(vm), _p(pm), _p(cam), C.c_int(W), C.c_int(H),
                         C.c_float(tanfovx), C.c_float(tanfovy), _p(st["radii"]), _p(st["means2D"]),
                         _p(st["depths"]), _p(st["cov3D"]), _p(st["rgb"]), _p(st["conic_opacity"]),
                         _p(st["clamped"]), _p(st["tiles_touched"]), _p(st["point_offsets"]))
    st["num_rendered"] = int(R)
    if "sort" not in stages:
        return st
    st["keys"] = np.zeros(max(R, 1), np.uint64)[:R]
    st["point_list"] = np.zeros(max(R, 1), np.uint32)[:R]
    st["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    keys = np.zeros(max(R, 1), np.uint64)
    plist = np.zeros(max(R, 1), np.uint32)
    L.gso_bin_sort(C.c_int(P), C.c_int(R), C.c_int(W), C.c_int(H), _p(st["radii"]), _p(st["means2D"]),
                   _p(st["depths"]), _p(st["point_offsets"]), _p(keys), _p(plist), _p(st["ranges"]))
    st["keys"], st["point_list"] = keys[:R], plist[:R]
    if "render" not in stages:
        return st
    feat = cp if cp is not None else st["rgb"]
    st["final_T"] = np.zeros((H, W), np.float32)
    st["n_contrib"] = np.zeros((H, W), np.uint32)
    st["out_color"] = np.zeros((3, H, W), np.float32)
    L.gso_render(C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(plist), _p(st["means2D"]), _p(feat),
                 _p(st["conic_opacity"]), _p(bgc), _p(st["final_T"]), _p(st["n_contrib"]), _p(st["out_color"]))
    st["_inputs"] = dict(means3D=m, shs=sh, colors_precomp=cp, scales=sc, rotations=rot, cov3D_precomp=c3p,
                         viewmatrix=vm, projmatrix=pm, campos=cam, bg=bgc, tanfovx=float(tanfovx),
                         tanfovy=float(tanfovy), scale_modifier=float(scale_modifier))
    return st


def backward(st, dL_dout_color):
    """Gradients for the forward described by `st` (as returned by forward())."""
    L = lib()
    i = st["_inputs"]
    P, W, H, M, D, R = st["P"], st["W"], st["H"], st["M"], st["D"], st["num_rendered"]
    dpix = _f(dL_dout_color)
    g = dict(
        dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 4), np.float32),
        dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
        dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
        dL_dsh=np.zeros((P, M, 3), np.float32), dL_dscales=np.zeros((P, 3), np.float32),
        dL_drotations=np.zeros((P, 4), np.float32))
    if P == 0:
        return g
    colors = i["colors_precomp"] if i["colors_precomp"] is not None else st["rgb"]
    plist = np.ascontiguousarray(st["point_list"])
    L.gso_render_backward(C.c_int(P), C.c_int(R), C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(plist), _p(i["bg"]),
                          _p(st["means2D"]), _p(st["conic_opacity"]), _p(colors), _p(st["final_T"]),
                          _p(st["n_contrib"]), _p(dpix), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]),
                          _p(g["dL_dopacity"]), _p(g["dL_dcolors"]))
    cov = i["cov3D_precomp"] if i["cov3D_precomp"] is not None else st["cov3D"]
    L.gso_preprocess_backward(C.c_int(P), C.c_int(D), C.c_int(M), _p(i["means3D"]), _p(st["radii"]), _p(i["shs"]),
                              _p(st["clamped"]), _p(i["scales"]), _p(i["rotations"]), C.c_float(i["scale_modifier"]),
                              _p(cov), _p(i["viewmatrix"]), _p(i["projmatrix"]), C.c_int(W), C.c_int(H),
                              C.c_float(i["tanfovx"]), C.c_float(i["tanfovy"]), _p(i["campos"]),
                              _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dcolors"]), _p(g["dL_dmeans3D"]),
                              _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def backward_f64(state, dL_dout_color):
    """The exact-arithmetic (float64) gradient for a float32 forward state.

    `state`: dict with the float32 forward's artefacts -- P, W, H, M, D, ranges [T,2], point_list [R], means2D [P,2],
    conic_opacity [P,4], colors [P,3] (the clamped rgb, or colors_precomp), clamped [P,3] u8, final_T [H,W],
    n_contrib [H,W], radii [P], cov3D [P,6] (computed or precomp), and the inputs means3D, shs (or None), scales /
    rotations (or None), viewmatrix, projmatrix, campos, bg, tanfovx, tanfovy, scale_modifier.  The discrete decisions
    (tile lists, last contributors) are the float32 run's; every sum and product behind the gradients is redone in
    binary64, so the result is what the float32 implementations -- the reference with its atomics in scheduling order,
    ours in a fixed order -- are both roundings of.  Returns float64 arrays named like backward()'s."""
    L = lib64()
    d = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    P, W, H, M, D = (int(state[k]) for k in ("P", "W", "H", "M", "D"))
    plist = np.ascontiguousarray(np.asarray(state["point_list"]).astype(np.uint32))
    R = int(plist.shape[0])
    ranges = np.ascontiguousarray(np.asarray(state["ranges"]).astype(np.uint32))
    ncon = np.ascontiguousarray(np.asarray(state["n_contrib"]).astype(np.uint32))
    radii = np.ascontiguousarray(np.asarray(state["radii"]).astype(np.int32))
    clamped = np.ascontiguousarray(np.asarray(state["clamped"]).astype(np.uint8))
    shs, sc, rot = d(state.get("shs")), d(state.get("scales")), d(state.get("rotations"))
    g = dict(dL_dmeans2D=np.zeros((P, 3)), dL_dconic=np.zeros((P, 4)), dL_dopacity=np.zeros((P, 1)),
             dL_dcolors=np.zeros((P, 3)), dL_dmeans3D=np.zeros((P, 3)), dL_dcov3D=np.zeros((P, 6)),
             dL_dsh=np.zeros((P, max(M, 1), 3))[:, :M], dL_dscales=np.zeros((P, 3)), dL_drotations=np.zeros((P, 4)))
    g["dL_dsh"] = np.ascontiguousarray(g["dL_dsh"])
    if P == 0:
        return g
    # The per-Gaussian arrays of a forward state are only defined on the VISIBLE rows (the reference leaves the rest of its
    # geometry chunk as the allocator handed it over: arbitrary bit patterns, signalling NaNs among them).  No list entry and
    # no radii > 0 row refers to them, but the float32 -> float64 cast would raise "invalid value" on them -- and a warning
    # that is always there would also hide a NaN that matters.  They are zeroed first, and the casts run with invalid = raise.
    invisible = radii <= 0

    def dv(a):
        a32 = np.array(a, dtype=np.float32, copy=True)
        a32[invisible] = 0.0
        with np.errstate(invalid="raise"):
            return np.ascontiguousarray(a32.astype(np.float64))
    bg, m2, co, col = d(state["bg"]), dv(state["means2D"]), dv(state["conic_opacity"]), dv(state["colors"])
    with np.errstate(invalid="raise"):
        fT, dpix = d(state["final_T"]), d(dL_dout_color)
    L.gso_render_backward(C.c_int(P), C.c_int(R), C.c_int(W), C.c_int(H), _p(ranges), _p(plist), _p(bg), _p(m2), _p(co),
                          _p(col), _p(fT), _p(ncon), _p(dpix), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]),
                          _p(g["dL_dopacity"]), _p(g["dL_dcolors"]))
    m3, vm, pm, cam = (d(state[k]) for k in ("means3D", "viewmatrix", "projmatrix", "campos"))
    cov = dv(state["cov3D"])
    L.gso_preprocess_backward(C.c_int(P), C.c_int(D), C.c_int(M), _p(m3), _p(radii), _p(shs), _p(clamped), _p(sc), _p(rot),
                              C.c_double(float(state.get("scale_modifier", 1.0))), _p(cov), _p(vm), _p(pm), C.c_int(W),
                              C.c_int(H), C.c_double(float(state["tanfovx"])), C.c_double(float(state["tanfovy"])),
                              _p(cam.reshape(-1)), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dcolors"]),
                              _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]),
                              _p(g["dL_drotations"]))
    return g

"""First-light check on an MI355X: ours vs the reference's own rasterizer
(oracle/_ref) and vs the C restatement, on seeded scenes.  Prints a report;
writes nothing outside gpurun_out/."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from frosting_amd import _lib, scenes
from frosting_amd.introspect import State
from oracle import ref_rasterizer as REF
import helpers as Hh


def cmp_int(name, a, b):
    a, b = a.cpu(), b.cpu()
    nd = int((a != b).sum())
    print(f"    {name:18s} mismatches {nd}/{a.numel()}")
    return nd


def cmp_f(name, a, b, mask=None):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    if mask is not None:
        a, b = a[mask], b[mask]
    d = (a - b).abs()
    bit = int((a != b).sum())
    print(f"    {name:18s} max|d| {float(d.max()) if d.numel() else 0:.3e} mean|d| {float(d.mean()) if d.numel() else 0:.3e} "
          f"rel_l2 {Hh.rel_l2(a, b):.3e} not-bit-equal {bit}/{a.numel()}")


def check(name, P, cfg, view=0, mode="sh", cov="sr", exact=1, bwd=True, variant="exact"):
    dev = torch.device("cuda:0")
    print(f"== {name}: P={P} cfg={cfg} view={view} mode={mode} cov={cov} exact_blend={exact} ref={variant}")
    scene, cam, bg = scenes.config_scene(cfg, view, P=P)
    _lib.set_option("exact_blend", exact)
    (R, color, radii, geom, binning, img), args = Hh.run_ours_native(scene, cam, bg, dev, mode, cov)
    torch.cuda.synchronize()
    st = State(P, cam.image_width, cam.image_height, R, geom, binning, img)
    kw = Hh.oracle_kwargs(scene, cam, bg, mode, cov, as_numpy=False, device=dev)
    Rr, rcolor, rradii, rst = REF.forward(**kw, variant=variant)
    vis = (rradii > 0).cpu()
    print(f"    num_rendered ours {R} ref {Rr}; visible {int(vis.sum())}; max tile {int(st.tile_count.max())}")
    cmp_int("radii", radii, rradii)
    cmp_int("tiles_touched", st.tiles_touched, rst.tiles_touched)
    cmp_int("point_offsets", st.point_offsets, rst.point_offsets)
    cmp_f("means2D", st.means2D, rst.means2D, vis)
    cmp_f("depths", st.depths, rst.depths, vis)
    cmp_f("conic_opacity", st.conic_opacity, rst.conic_opacity, vis)
    if mode == "sh":
        cmp_f("rgb", st.rgb, rst.rgb, vis)
    cmp_int("ranges", st.ranges, rst.ranges)
    if R == Rr and R > 0:
        cmp_int("point_list", st.point_list, rst.point_list)
        cmp_int("sort keys", st.sort_keys(), rst.point_list_keys)
    cmp_int("n_contrib", st.n_contrib, rst.n_contrib)
    cmp_f("final_T", st.final_T, rst.final_T)
    cmp_f("image", color, rcolor)
    if not bwd:
        return
    gpix, _ = scenes.l1_target_grad(color.cpu(), 7)
    gpix = gpix.to(dev)
    bargs = (args[0], args[1], radii, args[2], args[4], args[5], args[6], args[7], args[8], args[9], args[10], args[11],
             gpix, args[14], args[15], args[16], geom, R, binning, img, False)
    from frosting_amd.rasterizer import _C
    g = _C.rasterize_gaussians_backward(*bargs)
    torch.cuda.synchronize()
    g2 = _C.rasterize_gaussians_backward(*bargs)
    torch.cuda.synchronize()
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]
    det = all(torch.equal(a, b) for a, b in zip(g, g2))
    print(f"    backward bit-reproducible run-to-run: {det}")
    rg = REF.backward(rst, gpix)
    rg2 = REF.backward(rst, gpix)
    for n, a in zip(names, g):
        if a.numel() == 0:
            continue
        b = rg[n]
        print(f"    {n:14s} rel_l2 vs ref {Hh.rel_l2(a.cpu(), b.cpu()):.3e}   (ref vs ref rerun {Hh.rel_l2(rg2[n].cpu(), b.cpu()):.3e})  max|ref| {float(b.abs().max()):.3e}")


def timing(cfg, P=None, iters=5):
    dev = torch.device("cuda:0")
    scene, cam, bg = scenes.config_scene(cfg, 0, P=P)
    P = scene.P
    from frosting_amd.rasterizer import _C
    for exact in (0, 1):
        _lib.set_option("exact_blend", exact)
        (R, color, radii, geom, binning, img), args = Hh.run_ours_native(scene, cam, bg, dev)
        gpix, _ = scenes.l1_target_grad(color.cpu(), 7)
        gpix = gpix.to(dev)
        bargs = (args[0], args[1], radii, args[2], args[4], args[5], args[6], args[7], args[8], args[9], args[10],
                 args[11], gpix, args[14], args[15], args[16], geom, R, binning, img, False)
        tf, tb = [], []
        for it in range(iters):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = _C.rasterize_gaussians(*args)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            _C.rasterize_gaussians_backward(*bargs)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            tf.append(t1 - t0); tb.append(t2 - t1)
        print(f"  ours cfg={cfg} P={P} R={R} exact={exact}: fwd {1e3*np.median(tf):.3f} ms  bwd {1e3*np.median(tb):.3f} ms")
    _lib.set_option("exact_blend", 0)
    kw = Hh.oracle_kwargs(scene, cam, bg, as_numpy=False, device=dev)
    for variant in ("fast",):
        tf, tb = [], []
        for it in range(iters):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            Rr, rcolor, rradii, rst = REF.forward(**kw, variant=variant)
            t1 = time.perf_counter()
            REF.backward(rst, gpix)
            t2 = time.perf_counter()
            tf.append(t1 - t0); tb.append(t2 - t1)
        print(f"  reference({variant}) cfg={cfg} P={P} R={Rr}: fwd {1e3*np.median(tf):.3f} ms  bwd {1e3*np.median(tb):.3f} ms (includes its zero-fill + allocs)")


if __name__ == "__main__":
    print("torch", torch.__version__, "device", torch.cuda.get_device_name(0), "cpus", os.cpu_count())
    print("reference dir on box:", os.path.exists("/root/reference"))
    steps = [
        lambda: check("tiny", 2000, "c2", exact=1),
        lambda: check("tiny-fast", 2000, "c2", exact=0, bwd=False),
        lambda: check("c2-exact", 100_000, "c2", exact=1),
        lambda: check("c2-colors-cov", 100_000, "c2", mode="colors", cov="cov", exact=1),
        lambda: check("c2-fast-vs-fastref", 100_000, "c2", exact=0, variant="fast"),
        lambda: check("c3-300k", 300_000, "c3", exact=1),
        lambda: timing("c2"),
        lambda: timing("c3", P=300_000),
        lambda: check("c3-full", 3_000_000, "c3", exact=1),
        lambda: timing("c3"),
    ]
    for s in steps:
        try:
            s()
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()

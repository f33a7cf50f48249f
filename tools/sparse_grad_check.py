"""Where do the default-arithmetic gradients of sparse frames on a large image sit?  

For every frame: ours (default = fast arithmetic; EXACT) and four runs of the reference's
own backward (oracle/_ref, atomics in scheduling order) -- each against the other AND against the float64 gradient of the
same float32 forward state (oracle/_build/libgs_oracle_f64.so): the exact-arithmetic value all of them are roundings of.
usage: python tools/sparse_grad_check.py [--frames c3:50000,c3:150000,...]  [--truth-max-points N]
TEST / MEASUREMENT TOOL: imports oracle/ (never part of the product path)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from frosting_amd import _lib, scenes
from oracle import gs_oracle as G
from oracle import ref_rasterizer as REF
import helpers as Hh

NAMES = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]


def frame(spec):
    cfg, P = spec.split(":")
    P = int(P)
    if cfg == "c2big":      # the C2 scene rendered on the C3 image
        scene, _, bg = scenes.config_scene("c2", 0, P=P)
        _, cam, _ = scenes.config_scene("c3", 1, P=8)
        return scene, cam, bg
    return scenes.config_scene(cfg, 2, P=P)


def truth_state(rst, scene, cam, bg):
    c = lambda t: t.detach().cpu().numpy()
    return dict(P=rst.P, W=rst.W, H=rst.H, M=scene.shs.shape[1], D=scene.sh_degree, ranges=c(rst.ranges), point_list=c(rst.point_list),
                means2D=c(rst.means2D), conic_opacity=c(rst.conic_opacity), colors=c(rst.rgb), clamped=c(rst.clamped),
                final_T=c(rst.final_T), n_contrib=c(rst.n_contrib), radii=c(rst.radii), cov3D=c(rst.cov3D),
                means3D=scene.means3D.numpy(), shs=scene.shs.numpy(), scales=scene.scales.numpy(), rotations=scene.rotations.numpy(),
                viewmatrix=cam.viewmatrix.numpy(), projmatrix=cam.projmatrix.numpy(), campos=cam.campos.numpy(), bg=bg.numpy(),
                tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, scale_modifier=1.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="c3:50000,c3:150000,c3:400000,c2big:100000,c2:60000,c3:3000000")
    ap.add_argument("--truth-max-points", type=int, default=3_000_000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ops = Hh.native_ops("ext")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import _bwd_args
    for spec in a.frames.split(","):
        scene, cam, bg = frame(spec)
        _lib.set_option("exact_blend", 1)
        out, args = Hh.run_ours_native(scene, cam, bg, dev, ops=ops)
        _, rcolor, _, rst = REF.forward(**Hh.oracle_kwargs(scene, cam, bg, as_numpy=False, device=dev))
        assert torch.equal(out[1], rcolor)
        gpix, _ = scenes.l1_target_grad(out[1].cpu(), 9)
        gpix = gpix.to(dev)
        runs = Hh.reference_runs(lambda: REF.backward(rst, gpix))
        active = int((rst.ranges[:, 1] > rst.ranges[:, 0]).sum())
        print(f"\n=== {spec}: R = {out[0]}, non-empty tiles {active} of {rst.ranges.shape[0]}", flush=True)
        ours = {}
        ours["exact/auto"] = [g.clone() for g in ops.rasterize_gaussians_backward(*_bwd_args(args, out, gpix))]
        _lib.set_option("exact_blend", 0)
        out2, _ = Hh.run_ours_native(scene, cam, bg, dev, ops=ops)
        b2 = _bwd_args(args, out2, gpix)
        ours["fast/auto"] = [g.clone() for g in ops.rasterize_gaussians_backward(*b2)]
        truth = None
        if scene.P <= a.truth_max_points:
            truth = G.backward_f64(truth_state(rst, scene, cam, bg), gpix.cpu().numpy())
        hdr = f"    {'tensor':14s} {'ref~ref':>9s}" + "".join(f" {k:>16s}" for k in ours)
        print(hdr + ("   | vs float64:  ref(min..max)      " + "".join(f" {k:>16s}" for k in ours) if truth else ""))
        for i, name in enumerate(NAMES):
            noise = Hh.reference_noise(runs, name)
            line = f"    {name[3:]:14s} {noise:9.1e}" + "".join(f" {Hh.distance_to_reference(g[i], runs, name):16.1e}" for g in ours.values())
            if truth:
                t = truth[name]
                rr = [Hh.rel_l2(r[name].cpu().numpy(), t) for r in runs]
                line += f"   |              {min(rr):.1e}..{max(rr):.1e}  " + "".join(f" {Hh.rel_l2(g[i].cpu().numpy(), t):16.1e}" for g in ours.values())
            print(line, flush=True)
        if truth:
            # how concentrated is the float32 error?  per Gaussian: the reference's worst run against the float64 value
            for name in ("dL_dmeans3D", "dL_dscales", "dL_drotations"):
                i = NAMES.index(name)
                t = torch.from_numpy(truth[name]).reshape(scene.P, -1)
                e_ref = torch.stack([(r[name].cpu().double().reshape(scene.P, -1) - t).pow(2).sum(1) for r in runs]).max(0).values
                tn = t.pow(2).sum(1)
                tot = float(e_ref.sum())
                srt = torch.sort(e_ref, descending=True).values
                conc = [float(srt[:k].sum()) / max(tot, 1e-300) for k in (1, 10, 100, 1000)]
                rel = (e_ref / tn.clamp_min(1e-300)).sqrt()
                vis = tn > 0
                line = f"    {name[3:]:12s} share of the reference's squared error in its top 1/10/100/1000 Gaussians: " + " ".join(f"{c:.3f}" for c in conc)
                line += f" | Gaussians with ref rel. error > 1e-2: {int((rel[vis] > 1e-2).sum())} of {int(vis.sum())}, > 1e-3: {int((rel[vis] > 1e-3).sum())}"
                print(line)
                for frac in (1e-4, 1e-3):
                    k = max(1, int(frac * int(vis.sum())))
                    drop = torch.topk(e_ref, k).indices
                    keep = torch.ones(scene.P, dtype=torch.bool); keep[drop] = False
                    tk = t[keep]
                    f = lambda g: float((g.cpu().double().reshape(scene.P, -1)[keep] - tk).norm() / tk.norm())
                    rr = [f(r[name]) for r in runs]
                    print(f"      without the {k} worst-conditioned Gaussians (ref vs float64): ref {min(rr):.1e}..{max(rr):.1e} | " +
                          " ".join(f"{kk} {f(g[i]):.1e}" for kk, g in ours.items()), flush=True)
        del rst, runs, ours, out, out2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

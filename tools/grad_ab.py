"""Which ingredient of the default blend arithmetic is behind its distance to the float64 gradient?  

One process = one library build (FROSTING_LIB selects an A/B build of tools/build_variants.sh; the ctypes binding honours
it).  For every frame: the reference's forward state and four runs of its backward (oracle/_ref), the float64 gradient of
that state, and OURS in the default arithmetic -- judged on the float32-computable rows exactly as tests/helpers.py does
(well_ours / well_ref per tensor), without asserting.
usage: [FROSTING_LIB=...] python tools/grad_ab.py [--frames c3:400000,c2big:100000] [--exact]
TEST / MEASUREMENT TOOL: imports oracle/ (never part of the product path)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from frosting_amd import _lib, scenes
from oracle import ref_rasterizer as REF
import helpers as Hh
from sparse_grad_check import frame


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="c3:400000,c2big:100000")
    ap.add_argument("--exact", action="store_true", help="also print the EXACT arithmetic's row")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ops = Hh.native_ops("ctypes")
    from test_gpu_parity import _bwd_args
    print(f"library: {os.environ.get('FROSTING_LIB', '(default build)')}")
    for spec in a.frames.split(","):
        scene, cam, bg = frame(spec)
        _, rcolor, _, rst = REF.forward(**Hh.oracle_kwargs(scene, cam, bg, as_numpy=False, device=dev))
        gpix, _ = scenes.l1_target_grad(rcolor.cpu(), 9)
        gpix = gpix.to(dev)
        runs = Hh.reference_runs(lambda: REF.backward(rst, gpix))
        truth = Hh.truth_from_ref_state(rst, gpix)
        for exact in ((1, 0) if a.exact else (0,)):
            _lib.set_option("exact_blend", exact)
            out, args = Hh.run_ours_native(scene, cam, bg, dev, ops=ops)
            grads = ops.rasterize_gaussians_backward(*_bwd_args(args, out, gpix))
            rep = Hh.judge_gradients(grads, runs, truth, fast=not exact, label=spec, quiet=True, check=False)
            print(f"  {spec:16s} {'EXACT' if exact else 'default'}: " + "  ".join(
                f"{n[3:]} {r['well_ours']:.2e}/{r['well_ref']:.2e}={r['well_ours'] / max(r['well_ref'], 1e-300):.1f}x" for n, r in rep.items()), flush=True)
            # is ours' distance on the computable rows spread over the Gaussians, or does it sit in a few of them?  (the rows set
            # aside are the REFERENCE's worst; ours may draw badly on others)
            for name in ("dL_dmeans2D", "dL_dmeans3D", "dL_dcov3D"):
                t = Hh._rows(truth[name], scene.P, dev)
                o = Hh._rows(dict(zip(Hh.GRAD_NAMES, grads))[name], scene.P, dev)
                rs = [Hh._rows(r[name], scene.P, dev) for r in runs]
                e_ref = torch.stack([(r - t).pow(2).sum(1) for r in rs]).max(0).values
                live = int((t.pow(2).sum(1) > 0).sum())
                k = max(1, int(-(-Hh.TRIM_FRACTION * live // 1)))
                keep = torch.ones(scene.P, dtype=torch.bool, device=dev)
                keep[torch.topk(e_ref, k).indices] = False
                e_o = (o - t).pow(2).sum(1) * keep
                srt = torch.sort(e_o, descending=True).values
                tot = float(e_o.sum())
                tk = float(t[keep].norm())
                shares = " ".join(f"{float(srt[:m].sum()) / max(tot, 1e-300):.2f}" for m in (1, 10, 100, 1000))
                rest = [float((e_o.sum() - srt[:m].sum()).clamp_min(0).sqrt()) / tk for m in (10, 100, 1000)]
                print(f"      {name[3:]:9s} share of ours' squared distance in its worst 1/10/100/1000 Gaussians: {shares} | without them (10/100/1000): "
                      + " ".join(f"{x:.1e}" for x in rest), flush=True)
        del rst, runs, truth
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

"""Shared helpers for the parity tests: run ours / the C oracle / the reference on
the same seeded inputs."""
from __future__ import annotations

import numpy as np
import torch

from frosting_amd import scenes
from frosting_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, _C


def settings_for(cam, bg, sh_degree, device, scale_modifier=1.0, debug=False):
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=bg.to(device), scale_modifier=scale_modifier, viewmatrix=cam.viewmatrix.to(device),
        projmatrix=cam.projmatrix.to(device), sh_degree=sh_degree, campos=cam.campos.to(device),
        prefiltered=False, debug=debug)


def cov3d_from(scene):
    """Python-side covariance (strip_symmetric(L L^T), L = R S) as the reference's
    compute_cov3D_python path builds it (gaussian_splatting/utils/general_utils.py:64-110)."""
    q = scene.rotations.double()   # float64, rounded once: independent of the host's BLAS / SIMD
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)
    L = R * scene.scales.double()[:, None, :]
    S = (L[:, :, None, :] * L[:, None, :, :]).sum(-1)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1).float().contiguous()


def precomp_colors(scene):
    """Deterministic 'precomputed colour' input (float64 sigmoid of the DC coefficients, rounded once)."""
    return torch.sigmoid(scene.shs[:, 0, :].double()).float().contiguous()


def native_ops(binding: str):
    """'ctypes': frosting_amd.rasterizer._C (ctypes over the C ABI); 'ext': the compiled torch extension
    diff_gaussian_rasterization._C (setup.py).  Both end in the same HIP library."""
    if binding == "ext":
        import diff_gaussian_rasterization
        return diff_gaussian_rasterization._C
    return _C


# ---- gradient bars ------------------------------------------------------------------------------------------------
# The gate is SURVEY 7.1 step 5: rel. L2 <= 1e-4 per gradient tensor.  Ours is bit-reproducible; the reference sums nine
# float atomics per (pixel, Gaussian) pair in scheduling order and differs from ITSELF run to run.  Round 4 measured what
# that spread is made of (tools/sparse_grad_check.py, profiles/r04_sparse_grad_check.log): on every scene 84 - 100 % of
# the reference's squared error against the float64 gradient sits in ONE Gaussian -- a needle seen edge-on whose cov2D
# is held invertible only by the 0.3-pixel low-pass, so that backward.cu:197-271 multiplies the rounding of its pixel
# sums by ~1e6.  On such a Gaussian every float32 evaluation (the reference's included) is a draw; whole-tensor rel. L2
# then measures which draw one got (the reference against itself: 3e-7 ... 7e-2 on dL_drotations from session to
# session), not the arithmetic.  So the yardstick is the exact-arithmetic gradient: oracle/_build/libgs_oracle_f64.so
# (gs_oracle.c with every float a double) run on the float32 forward state, and every comparison is made twice:
#   * on the float32-COMPUTABLE Gaussians -- all but the ceil(1e-4 x #rows) rows on which the reference's own runs are
#     farthest from the float64 value -- ours must meet the gate against the float64 gradient (where the reference
#     itself does not -- the thin shell of C4 seen edge-on: 9.7e-5 -- at most 1.5 x the reference's distance in EXACT
#     arithmetic and 2 x in the default arithmetic, the reference measured THE WAY OURS IS: each of its runs on the rows
#     the OTHER runs leave, the worst of them (`well_ref_loo`, 5 - 10 % above `well_ref`).  Round 6, 37 GPU sessions /
#     repeats of C4's default-arithmetic leg: ours reads 0.90e-4 ... 1.5e-4 on dL_drotations WITH THE SAME BITS, by which
#     32 rows the reference's four runs happen to set aside (a few rows of the edge-on shell carry ours' error and are
#     among the reference's worst only in some runs), the reference 0.91e-4 ... 1.1e-4; the 1.5 x form failed once (ratio
#     1.61), profiles/r06_c4_gradient_repeats.log) AND stay
#     within a small factor of the reference's own distance to it: 3 x in EXACT arithmetic (the reference's operation
#     order: what is left is the order of the additions, two draws of the same rounding noise), 5 x in the default
#     arithmetic (exp2 of a pre-scaled quadratic form in fused multiply-adds: measured 1 - 3.4 x);
#   * on the whole tensor, the ill-conditioned rows included, ours must be within max(5 x the reference's run-to-run
#     spread, gate) of the nearest reference run, OR no farther from the float64 gradient than 10 x the reference's
#     own worst run of four: where float32 cannot compute a row, every evaluation is a draw from a heavy-tailed
#     distribution (two draws were seen 3.6 x apart) and ours must merely not be out of the reference's league at it -- a
#     wrong clamp, a NaN, a lost slot would be off by orders of magnitude.
# No per-scene factors: the bars of rounds 1-3 (floors x 2 ... x 8 by scene, a flat 5e-3 for one case) are gone.
# Round 5 pins the yardstick itself and tightens the rest:
#   * WELL_REF_MAX: on the computable rows the float64 gradient and the reference's own runs must AGREE (<= 2e-4 per
#     tensor; the worst ever measured is 1.0e-4, dL_drotations of C4's edge-on shell) -- asserted in every comparison, so a
#     regression of the float64 oracle cannot silently widen every bar at once;
#   * the whole-tensor clause's second arm is 3 x the reference's worst run where the tensor has no ill-conditioned rows
#     (the rows set aside do not dominate the reference's own error: all_ref <= 2 x well_ref) and 5 x where they do
#     (there every evaluation is a draw; two draws of the reference itself were seen 3.6 x apart, ours 3.7 x from it:
#     profiles/r04_pytest_gpu.log, "principal=+0.1,-0.1" dL_drotations) -- it was 10 x everywhere;
#   * WELL_FLOOR of the default arithmetic 3e-5 -> 2e-5 (measured worst that needs it: dL_dmeans2D 1.6e-5 at C3);
#   * regression guards: WELL_OURS_MAX[fast][name], 2.5 x the largest well_ours of profiles/r04_pytest_gpu.log over every
#     GPU test -- a 10 - 30 x regression on a tensor where the reference sits at 1e-7 ... 1e-6 no longer hides under the gate.
GATE = 1e-4
TRIM_FRACTION = 1e-4
WELL_REF_MAX = 2e-4
ABOVE_GATE_FACTOR = {False: 1.5, True: 2.0}  # [fast]: where the reference itself misses the gate, ours vs the reference's distance
WELL_FACTOR = {False: 3.0, True: 5.0}        # [fast]: ours vs the reference's own distance to float64, computable rows
WELL_FLOOR = {False: 1e-5, True: 2e-5}       # ... below which that factor is not asked for (the reference itself sits at 1e-7 ... 1e-5)
WHOLE_FACTOR = 3.0                           # whole tensor vs the reference's worst run, no ill-conditioned rows ...
WHOLE_FACTOR_ILL = 5.0                       # ... and where the rows set aside dominate the reference's own error
# 2.5 x the largest well_ours over all 144 GPU tests of round 4 (profiles/r04_pytest_gpu.log; printed there with two
# digits), by arithmetic and tensor.  A scene on which the reference itself is farther keeps the 1.5 x well_ref bar.
_R04_WELL_OURS = {
    False: dict(dL_dmeans2D=2.6e-6, dL_dcolors=6.7e-7, dL_dopacity=1.2e-6, dL_dmeans3D=4.9e-6, dL_dcov3D=2.6e-5, dL_dsh=6.8e-7,
                dL_dscales=4.7e-5, dL_drotations=1.0e-4),
    True: dict(dL_dmeans2D=1.6e-5, dL_dcolors=2.5e-6, dL_dopacity=1.5e-6, dL_dmeans3D=1.1e-5, dL_dcov3D=9.3e-6, dL_dsh=2.5e-6,
               dL_dscales=5.2e-5, dL_drotations=7.4e-5),
}
WELL_OURS_MAX = {fast: {k: 2.5 * v for k, v in d.items()} for fast, d in _R04_WELL_OURS.items()}
GRAD_NAMES = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]


def _rows(a, P, device="cpu"):
    """[P, -1] float64 view of a gradient tensor / array on `device` (the comparisons of a 3 M-Gaussian frame -- norms and
    top-k over 144 M-element tensors, five of them per name -- take a minute on the host and a second on the GPU)"""
    t = a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    t = t.to(device=device, dtype=torch.float64)
    return t.reshape(P, -1) if t.numel() else t.reshape(P, 0)


def truth_from_ref_state(rst, dL_dimage):
    """float64 gradient (oracle.gs_oracle.backward_f64) of the float32 forward state of a reference run (RefState)."""
    from oracle import gs_oracle as G
    c = lambda t: None if t is None else t.detach().cpu().numpy()
    i = rst.inputs
    pre = i["colors_precomp"]
    state = dict(P=rst.P, W=rst.W, H=rst.H, M=i["M"], D=i["sh_degree"], ranges=c(rst.ranges), point_list=c(rst.point_list),
                 means2D=c(rst.means2D), conic_opacity=c(rst.conic_opacity), colors=c(pre) if pre is not None else c(rst.rgb),
                 clamped=c(rst.clamped), final_T=c(rst.final_T), n_contrib=c(rst.n_contrib), radii=c(rst.radii),
                 cov3D=c(i["cov3D_precomp"]) if i["cov3D_precomp"] is not None else c(rst.cov3D),
                 means3D=c(i["means3D"]), shs=c(i["shs"]), scales=c(i["scales"]), rotations=c(i["rotations"]),
                 viewmatrix=c(i["viewmatrix"]), projmatrix=c(i["projmatrix"]), campos=c(i["campos"]), bg=c(i["bg"]),
                 tanfovx=i["tanfovx"], tanfovy=i["tanfovy"], scale_modifier=i["scale_modifier"])
    return G.backward_f64(state, c(dL_dimage))


def truth_from_oracle_state(st, dL_dimage):
    """float64 gradient of the float32 forward state of the C oracle (gs_oracle.forward's dict)."""
    from oracle import gs_oracle as G
    i = st["_inputs"]
    pre = i["colors_precomp"]
    state = dict(P=st["P"], W=st["W"], H=st["H"], M=st["M"], D=st["D"], ranges=st["ranges"], point_list=st["point_list"],
                 means2D=st["means2D"], conic_opacity=st["conic_opacity"], colors=pre if pre is not None else st["rgb"],
                 clamped=st["clamped"], final_T=st["final_T"], n_contrib=st["n_contrib"], radii=st["radii"],
                 cov3D=i["cov3D_precomp"] if i["cov3D_precomp"] is not None else st["cov3D"], means3D=i["means3D"],
                 shs=i["shs"], scales=i["scales"], rotations=i["rotations"], viewmatrix=i["viewmatrix"],
                 projmatrix=i["projmatrix"], campos=i["campos"], bg=i["bg"], tanfovx=i["tanfovx"], tanfovy=i["tanfovy"],
                 scale_modifier=i["scale_modifier"])
    return G.backward_f64(state, np.asarray(dL_dimage))


def judge_gradients(ours, refs, truth, fast, label, names=None, quiet=False, guard=True, check=True):
    """ours: {name: tensor} (or the 8-tuple of rasterize_gaussians_backward); refs: list of {name: tensor / array} -- runs
    of a float32 reference on the same forward state (the reference's atomics, a stored fixture, the C oracle);
    truth: {name: float64 array}.  Asserts the two comparisons described above and returns the per-tensor report."""
    if not isinstance(ours, dict):
        ours = dict(zip(GRAD_NAMES, ours))
    dev = next((t.device for t in ours.values() if isinstance(t, torch.Tensor) and t.is_cuda), torch.device("cpu"))
    report = {}
    for name in (names or GRAD_NAMES):
        if name not in ours or name not in truth:
            continue
        t_full = np.asarray(truth[name])
        P = t_full.shape[0]
        t = _rows(t_full, P, dev)
        if t.numel() == 0 or not bool(t.any()):
            assert not bool(_rows(ours[name], P, dev).any()) or t.numel() == 0, (label, name, "expected an all-zero gradient")
            continue
        o = _rows(ours[name], P, dev)
        rs = [_rows(r[name], P, dev) for r in refs]
        e_ref = torch.stack([(r - t).pow(2).sum(1) for r in rs]).max(0).values          # per row: the reference's worst run
        live = int((t.pow(2).sum(1) > 0).sum())
        k = max(1, int(np.ceil(TRIM_FRACTION * live)))
        keep = torch.ones(P, dtype=torch.bool, device=dev)
        keep[torch.topk(e_ref, k).indices] = False
        tn, tkn = float(t.norm()), float(t[keep].norm())
        well_ours = float((o[keep] - t[keep]).norm()) / tkn
        well_ref = max(float((r[keep] - t[keep]).norm()) for r in rs) / tkn
        # the reference judged the way ours is: each run on the rows the OTHER runs leave (its own unlucky rows stay in)
        well_ref_loo = well_ref
        if len(rs) > 2:
            e_runs = torch.stack([(r - t).pow(2).sum(1) for r in rs])
            for i, r in enumerate(rs):
                others = torch.cat([e_runs[:i], e_runs[i + 1:]]).max(0).values
                keep_i = torch.ones(P, dtype=torch.bool, device=dev)
                keep_i[torch.topk(others, k).indices] = False
                well_ref_loo = max(well_ref_loo, float((r[keep_i] - t[keep_i]).norm()) / float(t[keep_i].norm()))
            del e_runs
        all_ours = float((o - t).norm()) / tn
        all_ref = max(float((r - t).norm()) for r in rs) / tn
        d_ref = min(float((o - r).norm()) / float(r.norm()) for r in rs)
        noise = max((float((rs[a] - rs[b]).norm()) / float(rs[b].norm()) for a in range(len(rs)) for b in range(a)), default=0.0)
        report[name] = dict(well_ours=well_ours, well_ref=well_ref, well_ref_loo=well_ref_loo, all_ours=all_ours, all_ref=all_ref, d_ref=d_ref, noise=noise, trimmed=k)
    if not quiet:
        print(f"\n[{label}, {'default' if fast else 'EXACT'} arithmetic] rel. L2 vs float64 on the float32-computable rows: ours (reference) | whole "
              f"tensor: ours vs float64 (reference vs float64), ours vs nearest reference run (reference vs itself)")
        for name, r in report.items():
            print(f"    {name[3:]:11s} {r['well_ours']:.1e} ({r['well_ref']:.1e}, one run on the others' rows {r['well_ref_loo']:.1e}; {r['trimmed']} rows set aside) | {r['all_ours']:.1e} "
                  f"({r['all_ref']:.1e}), {r['d_ref']:.1e} ({r['noise']:.1e})")
    if not check:          # (measurement tools: the numbers without the bars)
        return report
    for name, r in report.items():
        ratio = r["well_ours"] / max(r["well_ref"], 1e-300)
        msg = (label, "fast" if fast else "exact", name, f"ours / reference on the computable rows = {ratio:.2f}", r)
        # the yardstick: float64 and the reference's own runs agree on the float32-computable rows
        assert r["well_ref"] <= WELL_REF_MAX, ("the float64 yardstick and the reference disagree",) + msg
        assert r["well_ours"] <= max(GATE, ABOVE_GATE_FACTOR[fast] * r["well_ref_loo"]), msg
        assert r["well_ours"] <= max(WELL_FACTOR[fast] * r["well_ref"], WELL_FLOOR[fast]), msg
        cap = WELL_OURS_MAX[fast].get(name)
        assert guard is False or cap is None or r["well_ours"] <= max(cap, 1.5 * r["well_ref"]), ("regression guard (2.5 x the r04 record)",) + msg
        ill = r["all_ref"] > 2.0 * r["well_ref"]
        whole = WHOLE_FACTOR_ILL if ill else WHOLE_FACTOR
        assert r["d_ref"] < max(5.0 * r["noise"], GATE) or r["all_ours"] <= max(whole * r["all_ref"], GATE), msg
    return report


def reference_runs(backward_fn, n=4):
    """n runs of the reference's backward on the same state.  Its float atomics land in scheduling order, and on
    strongly cancelling sums (means3D / scales / quaternions of thin or anisotropic Gaussians) the run-to-run
    spread itself varies by an order of magnitude between pairs of runs (C4 means3D: 3e-5 .. 6e-4 measured), so
    ONE pair is not an estimate of it: a comparison that used one pair failed about one time in ten."""
    return [backward_fn() for _ in range(n)]


def reference_noise(runs, name):
    """largest pairwise rel. L2 difference between the reference's own runs"""
    return max((rel_l2(runs[i][name], runs[j][name]) for i in range(len(runs)) for j in range(i)), default=0.0)


def distance_to_reference(g, runs, name):
    """rel. L2 distance to the closest of the reference's runs"""
    return min(rel_l2(g, r[name]) for r in runs)


def run_ours_native(scene, cam, bg, device, mode="sh", cov="sr", debug=False, ops=None, scale_modifier=1.0, colors=None,
                    campos_2d=False):
    """Call the native entry points directly (what _RasterizeGaussians.forward does).
    colors: a [P,3] tensor handed over as colors_precomp (any per-Gaussian feature: Frosting renders view depth through
    this argument, frosting_model.py:1800-1811); campos_2d: the camera centre as a [1,3] tensor, which is what
    p3d_camera.get_camera_center() returns (frosting_model.py:1447,1462)."""
    sc = scene.to(device)
    e = torch.Tensor([])
    if colors is not None:
        mode = "colors"
    sh = sc.shs if mode == "sh" else e
    # colours are computed once on the CPU so every implementation sees the same bits
    col = e if mode == "sh" else (colors if colors is not None else precomp_colors(scene)).to(device)
    scales, rots = (sc.scales, sc.rotations) if cov == "sr" else (e, e)
    cov3 = e if cov == "sr" else cov3d_from(scene).to(device)
    campos = cam.campos.to(device)
    if campos_2d:
        campos = campos.view(1, 3)
    args = (bg.to(device), sc.means3D, col, sc.opacities, scales, rots, float(scale_modifier), cov3, cam.viewmatrix.to(device),
            cam.projmatrix.to(device), cam.tanfovx, cam.tanfovy, cam.image_height, cam.image_width, sh,
            sc.sh_degree, campos, False, debug)
    out = (ops or _C).rasterize_gaussians(*args)
    return out, args


def oracle_kwargs(scene, cam, bg, mode="sh", cov="sr", as_numpy=True, device=None, scale_modifier=1.0, colors=None):
    conv = (lambda t: t.numpy()) if as_numpy else (lambda t: t.to(device))
    kw = dict(means3D=conv(scene.means3D), opacities=conv(scene.opacities), viewmatrix=conv(cam.viewmatrix),
              projmatrix=conv(cam.projmatrix), campos=conv(cam.campos), bg=conv(bg), width=cam.image_width,
              height=cam.image_height, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, sh_degree=scene.sh_degree)
    if scale_modifier != 1.0:
        kw["scale_modifier"] = float(scale_modifier)
    if colors is not None:
        mode = "colors"
    if mode == "sh":
        kw["shs"] = conv(scene.shs)
    elif colors is not None:
        kw["colors_precomp"] = conv(colors)
    else:
        kw["colors_precomp"] = conv(precomp_colors(scene))
    if cov == "sr":
        kw["scales"], kw["rotations"] = conv(scene.scales), conv(scene.rotations)
    else:
        kw["cov3D_precomp"] = conv(cov3d_from(scene))
    return kw


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).flatten()
    b = torch.as_tensor(b, dtype=torch.float64).flatten()
    d = (a - b).norm()
    n = b.norm()
    return float(d / n) if n > 0 else float(d)

"""CPU tests of the oracle itself (no GPU): the C restatement against
 (a) golden fixtures produced by the REFERENCE's own rasterizer on an MI355X
     (tests/golden/g_*.npz, tools/make_golden.py),
 (b) golden SH vectors produced by importing the reference's Python
     (tests/golden/sh_eval.npz, tools/make_golden_sh.py),
 (c) an independent float64 autograd derivation (tests/torch_ref.py),
 (d) identities the reference itself defines (SH-in-kernel == eval_sh,
     cov3D-in-kernel == R S^2 R^T)."""
import glob
import os

import numpy as np
import pytest
import torch

from frosting_amd import scenes, sh
from oracle import gs_oracle as G

import helpers as Hh

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RASTER_FIXTURES = sorted(glob.glob(os.path.join(GOLD, "g_*.npz")))


def _oracle_for(fx):
    scene, cam, bg = scenes.config_scene(str(fx["cfg"]), int(fx["view"]), P=int(fx["P"]))
    kw = Hh.oracle_kwargs(scene, cam, bg, str(fx["mode"]), str(fx["cov"]))
    return scene, cam, bg, G.forward(**kw)


@pytest.mark.parametrize("path", RASTER_FIXTURES, ids=[os.path.basename(p) for p in RASTER_FIXTURES])
def test_oracle_matches_reference_fixture(path):
    fx = np.load(path)
    scene, cam, bg, st = _oracle_for(fx)
    vis = fx["radii"] > 0
    # integer artefacts: bit-exact
    assert st["num_rendered"] == int(fx["num_rendered"])
    np.testing.assert_array_equal(st["radii"], fx["radii"])
    np.testing.assert_array_equal(st["tiles_touched"], fx["tiles_touched"].astype(np.uint32))
    np.testing.assert_array_equal(st["ranges"].astype(np.int64), fx["ranges"].astype(np.int64))
    np.testing.assert_array_equal(st["point_list"].astype(np.int64), fx["point_list"].astype(np.int64))
    np.testing.assert_array_equal(st["keys"].astype(np.int64), fx["keys"].astype(np.int64))
    # per-Gaussian floats: same IEEE operation order => identical values
    np.testing.assert_array_equal(st["depths"][vis], fx["depths"][vis])
    np.testing.assert_array_equal(st["means2D"][vis], fx["means2D"][vis])
    np.testing.assert_array_equal(st["conic_opacity"][vis], fx["conic_opacity"][vis])
    if str(fx["mode"]) == "sh":
        np.testing.assert_array_equal(st["rgb"][vis], fx["rgb"][vis])
    # image: libm expf vs the GPU's -> tolerance (north_star: <= 1e-4 per-pixel L1)
    l1 = np.abs(st["out_color"] - fx["image"]).mean()
    assert l1 <= 1e-6, l1
    assert (st["n_contrib"].astype(np.int64) != fx["n_contrib"].astype(np.int64)).mean() < 1e-3
    # backward: the reference sums with atomics in arbitrary order -> tolerance
    gpix, _ = scenes.l1_target_grad(torch.from_numpy(fx["image"]), int(fx["loss_seed"]))
    g = G.backward(st, gpix.numpy())
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
              "dL_drotations"):
        ref = fx["grad_" + k]
        if ref.size == 0 or not np.any(ref):
            continue
        assert Hh.rel_l2(g[k], ref) < 2e-4, (k, Hh.rel_l2(g[k], ref))


def test_fixtures_present():
    assert len(RASTER_FIXTURES) >= 3, "golden fixtures from the reference are missing (tools/make_golden.py)"


def test_sh_helper_matches_reference_python():
    fx = np.load(os.path.join(GOLD, "sh_eval.npz"))
    dirs, coef = torch.from_numpy(fx["dirs"]), torch.from_numpy(fx["coef"])
    for deg in range(5):
        np.testing.assert_allclose(sh.eval_sh(deg, coef, dirs).numpy(), fx[f"eval_deg{deg}"], rtol=1e-12, atol=1e-12)
    rgb = torch.from_numpy(fx["rgb"])
    np.testing.assert_allclose(sh.RGB2SH(rgb).numpy(), fx["rgb2sh"], rtol=1e-14)
    np.testing.assert_allclose(sh.SH2RGB(rgb).numpy(), fx["sh2rgb"], rtol=1e-14)


def test_config1_plumbing_sh_deg0_l1_on_cpu():
    """BASELINE config 1: 1k random Gaussians, SH degree-0 colour + L1, CPU torch only."""
    fx = np.load(os.path.join(GOLD, "sh_eval.npz"))
    means, sh0 = torch.from_numpy(fx["c1_means"]), torch.from_numpy(fx["c1_sh0"])
    col = sh.points_rgb(means, sh0, torch.from_numpy(fx["c1_campos"]), 0)
    np.testing.assert_allclose(col.numpy(), fx["c1_rgb"], rtol=1e-6, atol=1e-7)
    l1 = torch.abs(col - 0.5).mean()
    assert abs(float(l1) - float(fx["c1_l1"])) < 1e-7


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_in_kernel_sh_equals_python_sh(deg):
    """The reference defines SH-in-rasterizer == eval_sh + 0.5 + clamp
    (gaussian_renderer/__init__.py:73-78)."""
    scene, cam, bg = scenes.config_scene("mini", 2, P=800)
    kw = Hh.oracle_kwargs(scene, cam, bg)
    kw["sh_degree"] = deg
    st = G.forward(**kw, stages=("preprocess",))
    vis = st["radii"] > 0
    col = sh.points_rgb(scene.means3D.double(), scene.shs.double(), cam.campos.double(), deg).numpy()
    np.testing.assert_allclose(st["rgb"][vis], col[vis], rtol=0, atol=2e-6)
    assert np.array_equal(st["clamped"][vis].astype(bool), (col[vis] <= 0) & (st["rgb"][vis] == 0))


def test_precomputed_inputs_equal_in_kernel_paths():
    """colors_precomp / cov3D_precomp must reproduce the in-kernel SH / covariance paths."""
    scene, cam, bg = scenes.config_scene("mini", 4, P=1500)
    base = G.forward(**Hh.oracle_kwargs(scene, cam, bg))
    kw = Hh.oracle_kwargs(scene, cam, bg, cov="cov")
    alt = G.forward(**kw)
    assert (alt["radii"] != base["radii"]).mean() < 5e-3  # python-side L L^T rounds differently
    assert np.abs(alt["out_color"] - base["out_color"]).mean() < 1e-5
    vis = base["radii"] > 0
    kw2 = Hh.oracle_kwargs(scene, cam, bg)
    del kw2["shs"]
    kw2["colors_precomp"] = base["rgb"].copy()
    alt2 = G.forward(**kw2)
    np.testing.assert_array_equal(alt2["radii"], base["radii"])
    np.testing.assert_array_equal(alt2["out_color"], base["out_color"])


def test_mark_visible_and_culling():
    scene, cam, bg = scenes.config_scene("mini", 0, P=500)
    m = scene.means3D.clone()
    m[:50] = cam.campos + 0.05  # right at the camera -> behind the 0.2 near plane
    vis = G.mark_visible(m.numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy())
    z = (torch.cat([m, torch.ones(500, 1)], 1) @ cam.viewmatrix)[:, 2]
    np.testing.assert_array_equal(vis, (z > 0.2).numpy())
    assert not vis[:50].any()


def test_empty_and_ragged_inputs():
    scene, cam, bg = scenes.config_scene("mini", 0, P=300)
    st0 = G.forward(**Hh.oracle_kwargs(scenes.Scene(*(t[:0] for t in (scene.means3D, scene.scales, scene.rotations,
                                                     scene.opacities, scene.shs)), 3), cam, bg))
    assert st0["num_rendered"] == 0 and not st0["out_color"].any()  # P == 0: zeros, background NOT applied
    cam2 = scenes.ring_camera(1, 150, 101, 120.0, 120.0)  # W, H not multiples of 16
    st = G.forward(**Hh.oracle_kwargs(scene, cam2, bg))
    assert st["out_color"].shape == (3, 101, 150) and np.isfinite(st["out_color"]).all()
    assert st["num_rendered"] == int(st["tiles_touched"].sum()) == len(st["point_list"])
    k = st["keys"]
    assert (k[1:] >= k[:-1]).all()
    r = st["ranges"].astype(np.int64)
    assert ((r[:, 1] - r[:, 0]) >= 0).all() and (r[:, 1] - r[:, 0]).sum() == st["num_rendered"]


def test_oracle_backward_matches_independent_autograd():
    """Analytic backward of the oracle vs float64 autograd of tests/torch_ref.py."""
    import torch_ref
    P = 40
    scene = scenes.make_scene(P, 77, log_scale=np.log(0.08))
    cam = scenes.ring_camera(0, 48, 32, 40.0, 40.0)
    bg = torch.tensor([0.2, 0.4, 0.1])
    kw = Hh.oracle_kwargs(scene, cam, bg)
    st = G.forward(**kw)
    params = [t.double().clone().requires_grad_(True) for t in
              (scene.means3D, scene.scales, scene.rotations, scene.opacities, scene.shs)]
    img = torch_ref.render(*params, cam, bg, 3)
    l1 = float((img.detach().float() - torch.from_numpy(st["out_color"])).abs().mean())
    assert l1 < 1e-5, l1
    g = torch.Generator().manual_seed(5)
    wts = torch.randn(img.shape, generator=g, dtype=torch.float64)
    (img * wts).sum().backward()
    og = G.backward(st, wts.float().numpy())
    for name, p in zip(("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dsh"), params):
        assert Hh.rel_l2(og[name], p.grad.numpy()) < 2e-3, (name, Hh.rel_l2(og[name], p.grad.numpy()))


def _f64_state(st, scene, cam, bg):
    """the float32 forward state of G.forward as the dict G.backward_f64 takes"""
    return dict(P=st["P"], W=st["W"], H=st["H"], M=st["M"], D=st["D"], ranges=st["ranges"], point_list=st["point_list"],
                means2D=st["means2D"], conic_opacity=st["conic_opacity"], colors=st["rgb"], clamped=st["clamped"],
                final_T=st["final_T"], n_contrib=st["n_contrib"], radii=st["radii"], cov3D=st["cov3D"],
                means3D=scene.means3D.numpy(), shs=scene.shs.numpy(), scales=scene.scales.numpy(),
                rotations=scene.rotations.numpy(), viewmatrix=cam.viewmatrix.numpy(), projmatrix=cam.projmatrix.numpy(),
                campos=cam.campos.numpy(), bg=bg.numpy(), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, scale_modifier=1.0)


def test_float64_backward_is_the_limit_of_the_float32_one():
    """oracle/_build/libgs_oracle_f64.so (gs_oracle.c with every float a double) is the yardstick of the sparse-frame
    gradient tests on the GPU: here it is pinned on both sides -- within float32 rounding of the float32 oracle backward
    on the same forward state, and on the independent float64 autograd derivation."""
    import torch_ref
    P = 40
    scene = scenes.make_scene(P, 77, log_scale=np.log(0.08))
    cam = scenes.ring_camera(0, 48, 32, 40.0, 40.0)
    bg = torch.tensor([0.2, 0.4, 0.1])
    st = G.forward(**Hh.oracle_kwargs(scene, cam, bg))
    g = torch.Generator().manual_seed(5)
    wts = torch.randn(3, 32, 48, generator=g, dtype=torch.float64)
    g32 = G.backward(st, wts.float().numpy())
    g64 = G.backward_f64(_f64_state(st, scene, cam, bg), wts.float().numpy())
    for name in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"):
        assert g64[name].dtype == np.float64
        assert Hh.rel_l2(g32[name], g64[name]) < 2e-5, (name, Hh.rel_l2(g32[name], g64[name]))
    params = [t.double().clone().requires_grad_(True) for t in
              (scene.means3D, scene.scales, scene.rotations, scene.opacities, scene.shs)]
    img = torch_ref.render(*params, cam, bg, 3)
    (img * wts.float().double()).sum().backward()
    for name, p in zip(("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dsh"), params):
        d64, d32 = Hh.rel_l2(g64[name], p.grad.numpy()), Hh.rel_l2(g32[name], p.grad.numpy())
        # (the autograd side renders from float64 parameters end to end, the oracle from the float32 forward's 2-D
        # state: what is left between them is that state's rounding, not the backward's)
        assert d64 < max(2.0 * d32, 1e-5), (name, d64, d32)


def test_pure_torch_alpha_blend_baseline_matches_the_oracle():
    """oracle/torch_blend.py (the pure-PyTorch CPU alpha-blend bench.py times as north_star's CPU baseline): image,
    final_T and n_contrib against the C oracle's blend on the same 2-D state, its autograd gradients against the oracle's
    analytic blend backward (the reference's dL_dmean2D carries the NDC factor W/2, H/2 and its conic.y term is half
    the derivative: backward.cu:536-554)."""
    from oracle import torch_blend as TB
    scene, cam, bg = scenes.config_scene("mini", 3, P=1200)
    st = G.forward(**Hh.oracle_kwargs(scene, cam, bg))
    W, H = cam.image_width, cam.image_height
    gpix, _ = scenes.l1_target_grad(torch.from_numpy(st["out_color"]), 13)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    res = TB.render(t(st["means2D"]), t(st["conic_opacity"]), t(st["rgb"]), t(st["ranges"].astype(np.int64)),
                    t(st["point_list"].astype(np.int64)), bg, W, H, dL_dimage=gpix, chunk=64)
    assert float((res["image"] - t(st["out_color"])).abs().mean()) < 1e-6
    assert float((res["final_T"] - t(st["final_T"])).abs().max()) < 1e-5
    assert (res["n_contrib"] != t(st["n_contrib"].astype(np.int64))).float().mean() < 1e-3   # (libm expf vs torch.exp at a threshold)
    og = G.backward(st, gpix.numpy())
    ndc = torch.tensor([0.5 * W, 0.5 * H])
    assert Hh.rel_l2(res["dL_dmeans2D"] * ndc, og["dL_dmeans2D"][:, :2]) < 1e-4
    assert Hh.rel_l2(res["dL_dcolors"], og["dL_dcolors"]) < 1e-5
    assert Hh.rel_l2(res["dL_dconic_opacity"][:, 3], og["dL_dopacity"][:, 0]) < 1e-4
    co = res["dL_dconic_opacity"]
    mine = torch.stack([co[:, 0], 0.5 * co[:, 1], co[:, 2]], 1)
    assert Hh.rel_l2(mine, og["dL_dconic"][:, [0, 1, 3]]) < 1e-4
    # a subset of the tiles (what bench.py samples) touches only those tiles
    part = TB.render(t(st["means2D"]), t(st["conic_opacity"]), t(st["rgb"]), t(st["ranges"].astype(np.int64)),
                     t(st["point_list"].astype(np.int64)), bg, W, H, tiles=[0, 5, 17], chunk=64)
    gx = (W + 15) // 16
    for tile in (0, 5, 17):
        y0, x0 = (tile // gx) * 16, (tile % gx) * 16
        assert torch.equal(part["image"][:, y0:y0 + 16, x0:x0 + 16], res["image"][:, y0:y0 + 16, x0:x0 + 16])

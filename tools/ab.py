"""A/B timing of library options on the C3 step (one process, one scene build).
usage: python tools/ab.py [--config c3] [--points N] [--steps 30] "bwd_batch=3" "bwd_batch=2" ...
Each positional argument is a comma-separated list of option=value applied through frg_set_option; prints the
per-stage hipEvent times, the wall ms per step and the rel. L2 difference of the gradient buffer to the first setting."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from frosting_amd import _lib, scenes
from frosting_amd.parallel import ViewParallelRasterizer

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--points", type=int, default=0)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--view", type=int, default=0)
ap.add_argument("--order", default="input", choices=["input", "yrow", "morton"],
                help="memory order of the Gaussians: as generated | by the 16-pixel screen row of the projected centre | Morton order of the pixel")
ap.add_argument("--scene", default="config", choices=["config", "skew"], help="skew: scenes.make_skew_scene (bench.py's skew_scene)")
ap.add_argument("--shrink", type=float, default=1.0, help="scale the scene about the origin (positions and sizes): < 1 covers fewer tiles")
ap.add_argument("--deferred", action="store_true", help="frg_forward_deferred (no host synchronisation inside the step)")
ap.add_argument("settings", nargs="*", default=[""])
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = scenes.CONFIGS[a.config]
scene, cam, bg = scenes.config_scene(a.config, a.view, P=a.points or cfg["P"])
if a.scene == "skew":
    scene = scenes.make_skew_scene(a.points or cfg["P"], cfg["seed"] + 77)
if a.shrink != 1.0:
    scene = scenes.Scene((scene.means3D * a.shrink).contiguous(), (scene.scales * a.shrink).contiguous(), scene.rotations, scene.opacities, scene.shs, scene.sh_degree)
if a.order != "input":
    # what spatial coherence of the caller's array would be worth to the binning stages (timing experiment)
    ph = torch.cat([scene.means3D.double(), torch.ones(scene.P, 1, dtype=torch.float64)], 1) @ cam.projmatrix.double()
    w = ph[:, 3:4].clamp_min(1e-6)
    px = ((ph[:, 0:1] / w + 1) * cam.image_width - 1) * 0.5
    py = ((ph[:, 1:2] / w + 1) * cam.image_height - 1) * 0.5
    tx = (px[:, 0] / 16).floor().clamp(-4, 4095).long() + 4
    ty = (py[:, 0] / 16).floor().clamp(-4, 4095).long() + 4
    if a.order == "yrow":
        key = ty * 8192 + tx
    else:
        key = torch.zeros_like(tx)
        for bit in range(12):
            key |= ((tx >> bit) & 1) << (2 * bit)
            key |= ((ty >> bit) & 1) << (2 * bit + 1)
    perm = torch.argsort(key, stable=True)
    scene = scenes.Scene(scene.means3D[perm].contiguous(), scene.scales[perm].contiguous(), scene.rotations[perm].contiguous(),
                         scene.opacities[perm].contiguous(), scene.shs[perm].contiguous(), scene.sh_degree)
vpr = ViewParallelRasterizer(scene.to(dev), dev, deferred_counters=a.deferred)
cam_d, bg_d = cam.to(dev), bg.to(dev)
img, radii = vpr.forward(cam_d, bg_d)
gpix, _ = scenes.l1_target_grad(img.cpu(), 1)
gpix = gpix.to(dev)
def step():
    vpr.forward(cam_d, bg_d); vpr.backward(gpix, 0)
    if a.deferred and not vpr.finish():
        vpr.forward(cam_d, bg_d, deferred=False); vpr.backward(gpix, 0)
for _ in range(100): step()
base = None
defaults = {}
for setting in a.settings:
    pairs = [kv.split("=") for kv in setting.split(",") if kv]
    timed_stage = None      # pseudo-option timedprof=K: hipEvents around stage K stay ON in the timed loop (what bench.py's roofline costs)
    for k, v in pairs:
        if k == "timedprof":
            timed_stage = int(v); continue
        old = _lib.set_option(k, int(v)); defaults.setdefault(k, old)
    for _ in range(10): step()
    _lib.set_option("profile", 0)
    if timed_stage is not None:
        _lib.set_option("profile", 1); _lib.set_option("profile_stage", timed_stage)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps): step()
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / a.steps
    _lib.set_option("profile_stage", -1)
    _lib.set_option("profile", 1); _lib.stage_times()
    for _ in range(10): step()
    torch.cuda.synchronize()
    st = _lib.stage_times(); _lib.set_option("profile", 0)
    flat = vpr.exchange.flat.clone()
    if setting == a.settings[0]:
        from frosting_amd.introspect import State
        st_ = State(scene.P, cam.image_width, cam.image_height, vpr.true_num_rendered, vpr.geom.buf, vpr.binning.buf, vpr.img.buf)
        print(f"R {vpr.true_num_rendered}, tiles with a list {int((st_.tile_count > 0).sum())}, pixels with a contributor: tiles {int((st_.n_contrib.view(-1) > 0).sum())} px", flush=True)
    if base is None: base = flat
    d = float((flat.double() - base.double()).norm() / base.double().norm())
    print(f"[{setting or 'default':32s}] {ms:.4f} ms/step | " + " ".join(f"{k} {v:.3f}" for k, v in st.items() if v > 0) + f" | sum {sum(v for v in st.values() if v > 0):.3f} | grad diff vs first {d:.2e}", flush=True)
    for k, v in defaults.items(): _lib.set_option(k, v)

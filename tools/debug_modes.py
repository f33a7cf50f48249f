"""Does a forward depend on what its scratch buffers held before?  Pre-fills the three arenas (and the outputs) with
byte patterns before every forward and compares every artefact; then the two-thread per-call-mode run with diagnostics."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from frosting_amd import _lib, scenes
from frosting_amd.parallel import ViewParallelRasterizer
from frosting_amd.introspect import State
from frosting_amd.rasterizer import _C

dev = torch.device("cuda:0")
scene, cam, bg = scenes.config_scene("c2", 5, P=30_000)
cam_d, bg_d = cam.to(dev), bg.to(dev)
for name, md in (("default", {}), ("tight", dict(tight_binning=1)), ("async2", dict(async_sh=2)), ("tight+async2", dict(tight_binning=1, async_sh=2)),
                 ("exact", dict(exact_blend=1))):
    for k, v in md.items():
        _lib.set_option(k, v)
    vpr = ViewParallelRasterizer(scene.to(dev), dev)
    ref = None
    for pat in (None, 0x00, 0xFF, 0x7F, "rand", 0xFF):
        if pat is not None:
            for a in (vpr.geom, vpr.binning, vpr.img, vpr.work):
                if a.buf.numel():
                    if pat == "rand":
                        a.buf.copy_(torch.randint(0, 256, (a.buf.numel(),), dtype=torch.uint8, device=dev))
                    else:
                        a.buf.fill_(pat)
            vpr.radii.fill_(-7)
            if vpr.out_color is not None:
                vpr.out_color.fill_(float("nan"))
        img, radii = vpr.forward(cam_d, bg_d)
        g, _ = scenes.l1_target_grad(img.cpu(), 3)
        vpr.backward(g.to(dev), 0)
        st = State(scene.P, cam.image_width, cam.image_height, vpr.true_num_rendered, vpr.geom.buf, vpr.binning.buf, vpr.img.buf)
        cur = dict(img=img.clone(), radii=radii.clone(), pl=st.point_list.clone(), ranges=st.ranges.clone(), grads=vpr.exchange.flat.clone(),
                   m2=vpr.dL_dmeans2D.clone())
        if ref is None:
            ref = cur
        else:
            bad = [k for k in cur if not torch.equal(cur[k], ref[k])]
            print(f"[{name}] pattern {pat}: {'identical' if not bad else 'DIFFERS in ' + str(bad)}", flush=True)
    for k in md:
        _lib.set_option(k, 0)

# two threads, per-call modes
sc = scene.to(dev)
e = torch.Tensor([])
args = (bg_d, sc.means3D, e, sc.opacities, sc.scales, sc.rotations, 1.0, e, cam_d.viewmatrix, cam_d.projmatrix, cam.tanfovx, cam.tanfovy,
        cam.image_height, cam.image_width, sc.shs, sc.sh_degree, cam_d.campos, False, False)
for label, configs in (("exact|tight+async", [dict(exact_blend=1, tight_binning=0, async_sh=0), dict(exact_blend=0, tight_binning=1, async_sh=2)]),
                       ("exact|tight", [dict(exact_blend=1, tight_binning=0, async_sh=0), dict(exact_blend=0, tight_binning=1, async_sh=0)]),
                       ("exact|async", [dict(exact_blend=1, tight_binning=0, async_sh=0), dict(exact_blend=0, tight_binning=0, async_sh=2)]),
                       ("default|default", [dict(), dict()])):
    want = []
    for md in configs:
        out = _C.rasterize_gaussians(*args, modes=md)
        st = State(scene.P, cam.image_width, cam.image_height, out[0], out[3], out[4], out[5])
        want.append(dict(img=out[1].clone(), radii=out[2].clone(), pl=st.point_list.clone(), ranges=st.ranges.clone()))
    errors = []

    def worker(which):
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            for it in range(16):
                out = _C.rasterize_gaussians(*args, modes=configs[which])
                st = State(scene.P, cam.image_width, cam.image_height, out[0], out[3], out[4], out[5])
                cur = dict(img=out[1], radii=out[2], pl=st.point_list, ranges=st.ranges)
                torch.cuda.current_stream().synchronize()
                bad = [k for k in cur if not torch.equal(cur[k], want[which][k])]
                if bad:
                    d = float((cur["img"] - want[which]["img"]).abs().max()) if "img" in bad else 0.0
                    errors.append((which, it, bad, d))
    ts = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    print(f"[threads {label}] {'ok' if not errors else errors}", flush=True)

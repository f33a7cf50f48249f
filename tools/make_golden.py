"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref, exact build)
on an MI355X.  Run through gpurun; outputs land in gpurun_out/golden/ and are then
copied (small files) into tests/golden/ and committed.  Inputs are the seeded
synthetic scenes of frosting_amd.scenes, so fixtures store only outputs + the
recipe (P, config, view, mode, seed of the loss target)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from frosting_amd import scenes
from oracle import ref_rasterizer as REF
import helpers as Hh

CASES = [  # name, P, cfg, view, mode, cov   (small images keep the committed files small)
    ("g_sh_sr_3k", 3000, "mini", 0, "sh", "sr"),
    ("g_sh_sr_1k_v3", 1000, "mini", 3, "sh", "sr"),
    ("g_col_cov_2k", 2000, "mini", 1, "colors", "cov"),
    ("g_sh_cov_500_v5", 500, "mini", 5, "sh", "cov"),
]

def main():
    dev = torch.device("cuda:0")
    out_dir = os.path.join(ROOT, "gpurun_out", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, P, cfg, view, mode, cov in CASES:
        scene, cam, bg = scenes.config_scene(cfg, view, P=P)
        kw = Hh.oracle_kwargs(scene, cam, bg, mode, cov, as_numpy=False, device=dev)
        R, color, radii, st = REF.forward(**kw, variant="exact")
        gpix, _ = scenes.l1_target_grad(color.cpu(), 11)
        g = REF.backward(st, gpix.to(dev))
        vis = (radii > 0)
        arrs = dict(
            P=P, cfg=cfg, view=view, mode=mode, cov=cov, loss_seed=11, num_rendered=R,
            radii=radii.cpu().numpy(), tiles_touched=st.tiles_touched.cpu().numpy(),
            depths=st.depths.cpu().numpy(), means2D=st.means2D.cpu().numpy(),
            conic_opacity=st.conic_opacity.cpu().numpy(), rgb=st.rgb.cpu().numpy(),
            point_list=st.point_list.cpu().numpy(), keys=st.point_list_keys.cpu().numpy(),
            ranges=st.ranges.cpu().numpy(), n_contrib=st.n_contrib.cpu().numpy().astype(np.uint16),
            image=color.cpu().numpy().astype(np.float32),
        )
        for k, v in g.items():
            arrs["grad_" + k] = v.cpu().numpy()
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **arrs)
        print(name, "R", R, "visible", int(vis.sum()), "bytes", os.path.getsize(os.path.join(out_dir, name + ".npz")))

if __name__ == "__main__":
    main()

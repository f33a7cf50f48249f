#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Gaussian-splat rasterizer.

Metric (BASELINE.json): train views/sec, forward + backward rasterization at
3M Gaussians, 1600x1056, SH degree 3 (configs[2], "c3"), synthetic seeded scene
(SURVEY.md section 8d), inputs resident in HBM before the timed region.

One "step" = one pass of the hot path over one view: frg_forward + frg_backward
through the C ABI (the loss gradient dL/dimage is a fixed tensor; the loss itself
is outside the op and outside the byte model).  With --gpus N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank renders its own camera of
the 8-camera ring (view-parallel, weak scaling) and the per-Gaussian parameter
gradients are summed across ranks with one RCCL all-reduce per step -- the
exchange step north_star names.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from frosting_amd import _lib, scenes  # noqa: E402
from frosting_amd.parallel import ViewParallelRasterizer  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def stage_bytes(P, V, R, N, T):
    """Algorithmic HBM bytes per stage and per view (SURVEY.md 8(d) table;
    P Gaussians, V visible, R instances, N pixels, T tiles, 16 SH coefficients)."""
    return {
        "preprocess": 20 * P + 291 * V,
        "scan": 0,
        "scatter": 8 * P + 20 * V + 12 * R,
        "sort": 24 * R + 8 * R + 8 * T,
        "blend_fwd": 40 * R + 20 * N,
        "blend_bwd": 76 * R + 20 * N,
        "preprocess_bwd": 303 * V + 284 * P,
    }


def pmc_traffic(stage: str, cfg_name: str, P: int):
    """HBM bytes per launch of `stage` from the committed rocprofv3 PMC passes of this very
    workload (profiles/r01_pmc_traffic.json; counters cannot be read from inside the process).
    None when the run is not the profiled configuration."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if cfg_name != "c3" or P != scenes.CONFIGS["c3"]["P"] or not os.path.exists(path):
        return None
    try:
        return json.load(open(path))["per_launch"][stage]["hbm_bytes_corrected"]
    except Exception:
        return None


def cpu_baseline(cfg_name: str, P: int):
    """C restatement of the reference (oracle/gs_oracle.c, OpenMP) timed on the host
    cores for ONE forward+backward of the same workload -- reported, not a target."""
    from oracle import gs_oracle as G
    G.build()
    scene, cam, bg = scenes.config_scene(cfg_name, 0, P=P)
    kw = dict(means3D=scene.means3D.numpy(), opacities=scene.opacities.numpy(), viewmatrix=cam.viewmatrix.numpy(),
              projmatrix=cam.projmatrix.numpy(), campos=cam.campos.numpy(), bg=bg.numpy(), width=cam.image_width,
              height=cam.image_height, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, shs=scene.shs.numpy(),
              scales=scene.scales.numpy(), rotations=scene.rotations.numpy(), sh_degree=scene.sh_degree)
    t0 = time.perf_counter()
    st = G.forward(**kw)
    gpix = (np.sign(st["out_color"] - 0.5) / st["out_color"].size).astype(np.float32)
    G.backward(st, gpix)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "views/s", "cores": G.num_threads(), "kind": "port",
            "sample": f"1 view fwd+bwd of {cfg_name} at P={P} (R={st['num_rendered']}), {dt:.2f} s wall, OpenMP C port "
                      f"of the reference algorithm (oracle/gs_oracle.c)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spinup-steps", type=int, default=100,
                    help="untimed steps before the W warm-up steps: the process spends seconds on the host building the "
                         "scene, the idle GPU drops its clocks, and a short warm-up can end before they are back up "
                         "(seen once: 5.1 ms/step in the timed region against 2.05 in every other run)")
    ap.add_argument("--config", default="c3", choices=list(scenes.CONFIGS))
    ap.add_argument("--points", type=int, default=0, help="override the number of Gaussians (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact", action="store_true", help="use the EXACT blend arithmetic")
    ap.add_argument("--no-stage-timers", action="store_true", help="do not record per-stage hipEvents (no roofline object)")
    ap.add_argument("--tight-binning", action="store_true",
                    help="time the whole run with frg_set_option('tight_binning', 1): instances that cannot reach alpha >= "
                         "1/255 anywhere in their tile are dropped when the tile lists are built (outputs bit-identical, "
                         "lists = order-preserving sub-lists of the reference's).  Without the flag the headline run keeps "
                         "the reference's exact lists and the tight mode is timed in an extra pass (field 'tight_binning')")
    ap.add_argument("--no-tight-pass", action="store_true",
                    help="skip the informative extra pass that times tight binning after the timed region (used under "
                         "rocprofv3 so that per-kernel averages are not a mix of the two modes)")
    ap.add_argument("--deferred-counters", action="store_true",
                    help="use frg_forward_deferred (no host synchronisation inside the step) instead of frg_forward, "
                         "which like the reference blocks on a read-back of num_rendered; measured equal at C3")
    ap.add_argument("--exchange", default="factored", choices=["factored", "allreduce"],
                    help="N>1: 'allreduce' = one all-reduce of all 59 floats per Gaussian; 'factored' = all-reduce of the "
                         "11 non-SH floats + all-gather of the 3-float colour gradient, summed SH gradient rebuilt on "
                         "every rank (frosting_amd/parallel.py)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo only for functional tests of the multi-rank path on one GPU")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the exchange path on a single-rank RCCL group when launched without torch.distributed.run")
    ap.add_argument("--sync-exchange", action="store_true",
                    help="N>1: wait for the gradient all-reduce at the end of every step (no overlap with the next render)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FRG_BENCH_ONE_GPU"):   # functional test: every rank on GPU 0
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the rasterizer has no CPU path")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or args.force_exchange:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    cfg = scenes.CONFIGS[args.config]
    P = args.points or cfg["P"]
    scene, cam, bg = scenes.config_scene(args.config, rank % 8, P=P)
    _lib.set_option("exact_blend", 1 if args.exact else 0)
    _lib.set_option("profile", 0 if args.no_stage_timers else 1)
    _lib.set_option("tight_binning", 1 if args.tight_binning else 0)
    vpr = ViewParallelRasterizer(scene.to(dev), dev, process_group=dist.group.WORLD if dist else None,
                                 factor_sh=(args.exchange == "factored"), deferred_counters=args.deferred_counters)
    exchanging = dist is not None
    cam_d = cam.to(dev)
    bg_d = bg.to(dev)

    image, radii = vpr.forward(cam_d, bg_d)
    gpix, _ = scenes.l1_target_grad(image.cpu(), 20241022 + rank)
    gpix = gpix.to(dev)

    # Software-pipelined exchange (N > 1): the all-reduce of step k is enqueued right behind its
    # backward and only waited for when its gradient buffer is needed again (two buffers), so it
    # overlaps the render of step k+1.  Every step still performs its full forward, backward and
    # all-reduce, and all of them have completed when the timer stops.  --sync-exchange waits at
    # the end of every step instead.
    counter = [0]

    def step():
        slot = counter[0] % 2
        counter[0] += 1
        vpr.forward(cam_d, bg_d)
        if exchanging and not args.sync_exchange:
            # the exchange launched two steps ago on this buffer: its collectives are waited for here, its
            # SH rebuild runs on a side stream under the backward below
            vpr.prefetch_exchange(slot)
        vpr.backward(gpix, slot)             # writes straight into the flat gradient buffer
        if not vpr.finish():                 # deferred counters: more instances than the arena holds -> redo
            vpr.forward(cam_d, bg_d, deferred=False)
            vpr.backward(gpix, slot)
        if exchanging and not args.sync_exchange:
            vpr.wait_exchange(slot)          # join the rebuild before this buffer's collectives start again
        if exchanging:
            vpr.start_exchange(slot)
            if args.sync_exchange:
                vpr.wait_exchange(slot)

    def drain():
        if exchanging:
            vpr.wait_exchange(0)
            vpr.wait_exchange(1)

    timers = not args.no_stage_timers
    for _ in range(max(0, args.spinup_steps)):
        step()
    drain()
    for w in range(args.warmup):
        if timers and w == args.warmup - 1:
            drain()
            _lib.stage_times()               # forget the first steps (one-time stream / attribute set-up)
        step()
    drain()
    dom_stage = None
    if timers:
        # the warm-up ran with events around every stage: the dominant kernel is the one timed live
        # below (28 event records per step cost ~0.1 ms of stream time, 2 cost nothing measurable)
        warm = {k: v for k, v in _lib.stage_times().items() if v > 0}
        dom_stage = max(warm, key=warm.get) if warm else "blend_bwd"
        _lib.set_option("profile_stage", _lib.STAGE_NAMES.index(dom_stage))
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    stage_avg, dom_ms = {}, None
    if timers:
        # hipEvents recorded by the C ABI on the launch stream around the dominant kernel of every
        # timed step, read only now: its average launch duration over the timed region
        dom_ms = _lib.stage_times().get(dom_stage)
        # every stage once more, outside the timed region (informative: stage_ms)
        _lib.set_option("profile_stage", -1)
        for _ in range(5):
            step()
        drain()
        torch.cuda.synchronize(dev)
        stage_avg = {k: v for k, v in _lib.stage_times().items() if v > 0}
    tight = None
    if world == 1 and not exchanging and not args.tight_binning and not args.no_tight_pass:
        # informative extra pass, after and outside the timed region: the same steps with tight binning
        _lib.set_option("profile", 0)
        _lib.set_option("tight_binning", 1)
        for _ in range(3):
            step()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        dt_t = time.perf_counter() - t1
        _lib.set_option("tight_binning", 0)
        tight = {"ms_per_step": 1e3 * dt_t / args.steps, "value": args.steps / dt_t, "unit": "views/s",
                 "note": "option tight_binning=1 (off in the headline run): tile lists without the instances that provably "
                         "touch no pixel of their tile; image, radii, num_rendered and gradients bit-identical"}
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        views_per_s = world * args.steps / dt
        V = int((radii > 0).sum())
        R = int(vpr.true_num_rendered)
        N = cam.image_width * cam.image_height
        T = ((cam.image_width + 15) // 16) * ((cam.image_height + 15) // 16)
        from frosting_amd.introspect import State      # report-only: per-tile list statistics
        if not args.tight_binning and tight is not None:
            vpr.forward(cam_d, bg_d, deferred=False)   # the extra pass left tight lists in the buffers
        rng = State(P, cam.image_width, cam.image_height, R, vpr.geom.buf, vpr.binning.buf, vpr.img.buf).ranges
        tile_len = (rng[:, 1] - rng[:, 0]).float()
        B = stage_bytes(P, V, R, N, T)
        total_bytes = 312 * P + 614 * V + 160 * R + 40 * N + 8 * T
        out = {
            "metric": "train views/sec (fwd+bwd raster)", "value": views_per_s, "unit": "views/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {P} Gaussians, SH deg 3, {cam.image_width}x{cam.image_height}, "
                                   f"forward+backward, 1 view per GPU per step", "P": P, "visible": V,
                       "num_rendered": R, "instances_per_visible": R / max(V, 1), "tiles": T,
                       "tile_list_mean": float(tile_len.mean()), "tile_list_max": int(tile_len.max()),
                       "parallelism": f"view-parallel x{world}",
                       "exchange": ("none" if not exchanging else
                                    ("all-reduce of 59 floats/Gaussian" if args.exchange == "allreduce" else
                                     "factored: all-reduce of 11 floats/Gaussian + all-gather of dRGB (3 floats), "
                                     "summed SH gradient rebuilt per rank") +
                                    (", synchronous" if args.sync_exchange else
                                     ", overlapped with the next step's render (2 gradient buffers)")),
                       "exchange_bytes_per_rank": (4 * vpr.exchange.wire_floats_per_rank if exchanging else 0),
                       "blend_arithmetic": "exact" if args.exact else "fast", "seed": cfg["seed"],
                       "binning": "tight" if args.tight_binning else "reference-identical tile lists",
                       "counters": "deferred (no host synchronisation inside the step)" if args.deferred_counters
                                   else "blocking 48-byte read-back per forward"},
            "op_hbm": {"algorithmic_bytes_per_view": total_bytes,
                       "achieved_GBps_per_gpu": total_bytes / (dt / args.steps) / 1e9,
                       "frac_of_8TBps": total_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
        }
        if dom_ms and dom_ms > 0:
            dom = dom_stage
            ach = B[dom] / (dom_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, args.config, P),
                               "algorithmic_bytes_per_launch": B[dom], "avg_launch_ms": dom_ms,
                               "timed": "hipEvents around this kernel in every timed step"}
            out["stage_ms"] = stage_avg
            out["stage_ms_note"] = "all stages, 5 extra steps after the timed region"
        if tight:
            out["tight_binning"] = tight
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args.config, P)
            except Exception as ex:  # the baseline is informative; never lose the GPU number over it
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

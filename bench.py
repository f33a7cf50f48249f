#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Gaussian-splat rasterizer.

Metric (BASELINE.json): train views/sec, forward + backward rasterization at
3M Gaussians, 1600x1056, SH degree 3 (configs[2], "c3"), synthetic seeded scene
(SURVEY.md section 8d), inputs resident in HBM before the timed region.

One "step" = one pass of the hot path over one view: frg_forward + frg_backward
through the C ABI (the loss gradient dL/dimage is a fixed tensor; the loss itself
is outside the op and outside the byte model).  --config c2 times the forward only
(BASELINE configs[1]); --config c4 times the Frosting refine step (configs[3]: triangle
occlusion raster -> visible-face mask -> culled forward -> backward).

--gpus N > 1: one rank per GPU.  Launched as the driver does it (torch.distributed.run sets
WORLD_SIZE) the process is one rank; launched bare (`python bench.py --gpus N`) it re-executes
itself under torch.distributed.run with N ranks on 127.0.0.1.  Every rank renders its own camera
of the 8-camera ring (view-parallel, weak scaling) and the per-Gaussian parameter gradients are
summed across ranks over RCCL every step -- the exchange step north_star names.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spinup-steps", type=int, default=100,
                    help="untimed steps before the W warm-up steps: the process spends seconds on the host building the "
                         "scene, the idle GPU drops its clocks, and a short warm-up can end before they are back up")
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c4", "mini"])
    ap.add_argument("--views", type=int, default=8,
                    help="cameras of the 8-camera ring the timed loop cycles through (step i of rank r renders view (r + i) %% views "
                         "-- a trainer draws a new camera per step: refine.py:464-571); 1 = one fixed camera (round 1-3's loop, "
                         "still timed after the timed region as 'fixed_view')")
    ap.add_argument("--points", type=int, default=0, help="override the number of Gaussians (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the informative passes after the timed region (tight binning, API path, reference on this GPU, "
                         "skewed scene); used under rocprofv3 so that per-kernel averages are those of the headline mode only")
    ap.add_argument("--exact", action="store_true", help="use the EXACT blend arithmetic")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="frg_set_option(NAME, VALUE) before anything runs (tuning experiments; repeatable)")
    ap.add_argument("--no-stage-timers", action="store_true", help="do not record per-stage hipEvents (no roofline object)")
    ap.add_argument("--keep-for-backward", action="store_true",
                    help="C2 (forward only): the plain forward, which leaves checkpoints and work items for a backward, instead of "
                         "frg_forward_args::forward_only (A/B)")
    ap.add_argument("--composed-mask", action="store_true",
                    help="C4: the occlusion mask as the composition of torch.ones / cat / matmul, frg_mesh_visible_faces and a torch "
                         "index (rounds 2-4) instead of the one-call frg_mesh_occlusion_mask (A/B)")
    ap.add_argument("--tight-binning", action="store_true",
                    help="time the whole run with frg_set_option('tight_binning', 1): instances that cannot reach alpha >= "
                         "1/255 anywhere in their tile are dropped when the tile lists are built (outputs bit-identical, "
                         "lists = order-preserving sub-lists of the reference's).  Without the flag the headline run keeps "
                         "the reference's exact lists and the tight mode is timed in an extra pass (field 'tight_binning')")
    ap.add_argument("--no-tight-pass", action="store_true", help="(kept for older scripts) same as --no-extras")
    ap.add_argument("--deferred-counters", action="store_true",
                    help="use frg_forward_deferred (no host synchronisation inside the step) instead of frg_forward, "
                         "which like the reference blocks on a read-back of num_rendered")
    ap.add_argument("--exchange", default="slotsum", choices=["slotsum", "auto", "sparse", "factored", "allreduce"],
                    help="N>1: 'slotsum' (default, round 6) = the ranks all-gather the nine per-Gaussian SLOT SUMS of their view's "
                         "backward (48-byte rows of the Gaussians with a gradient, index-ordered behind a bit mask, fixed-capacity "
                         "packets: no host wait for a count) and every rank runs the per-Gaussian chain for every view's rows in "
                         "view order in ONE pass that writes each gradient row once (csrc/slot_exchange.hip) -- the plan the "
                         "arithmetic of DESIGN.md section 5 puts first at every bus bandwidth, and ONE kind of collective "
                         "(all_gather_into_tensor); 'allreduce' = all 59 floats per Gaussian are summed across ranks; 'factored' = the 11 non-SH "
                         "floats are summed + all-gather of the 3-float colour gradient, summed SH gradient rebuilt on "
                         "every rank; 'sparse' = only the ROWS of the Gaussians with a gradient travel -- (index, 11 floats, "
                         "dRGB), one visible Gaussian in seven at C3: counts, one padded all-gather, scatter-add in view order, "
                         "SH rebuild (frosting_amd/parallel.py); 'auto' = an explicit PROBE: N>1 on RCCL, 'slotsum', 'factored' and "
                         "'sparse' are each timed for a few steps before the warm-up and the fastest one runs (never the default: "
                         "the first multi-rank run of a build should execute as few never-executed collectives as possible)")
    ap.add_argument("--chunks", type=int, default=2,
                    help="slotsum: index ranges of Gaussians with a packet and a collective each -- the combine pass of one range "
                         "runs while the next range's packets travel.  2: single-rank exposed cost 0.146 - 0.16 ms and 6.06 x at 8 GPUs / "
                         "450 GB/s by the arithmetic; 4: 0.165 - 0.18 ms and 6.17 x (more overlap, more launches and collectives per step)")
    ap.add_argument("--phase1-in-pieces", action="store_true",
                    help="slotsum with several chunks: the backward's phase 1 chunk by chunk, every chunk's packet leaving as soon as its "
                         "sums exist (frg_backward_args::range_first / range_count) instead of one phase-1 call followed by all the packets")
    ap.add_argument("--reduce", default="allreduce", choices=["auto", "allreduce", "direct"],
                    help="N>1: how the summed part travels: 'allreduce' = one RCCL all-reduce; 'direct' = one RCCL reduce-scatter "
                         "of 1/N shards + one all-gather (every GPU talks to every other over its own xGMI link: SURVEY 8(e)); "
                         "'auto' = an explicit PROBE: both are timed on this run's buffers before the warm-up (RCCL backend, N>1) and "
                         "the faster one is used.  Only the 'factored' / 'allreduce' plans sum anything; default: allreduce")
    ap.add_argument("--probe-exchange", action="store_true",
                    help="run the 'auto' probe of --exchange on any backend (it is skipped on gloo otherwise: functional tests)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo only for functional tests of the multi-rank path on one GPU")
    ap.add_argument("--dry-run-plumbing", action="store_true",
                    help="no GPU: the launch path only -- self-launch under torch.distributed.run, rank environment, gloo group, the "
                         "slot-sum exchange's collectives (all_gather_into_tensor per chunk, verdict, capacity update) on toy packets, "
                         "barrier + max-over-ranks timing, rank 0's JSON line (marked dry_run; its value is NOT a measurement).  "
                         "tests/test_parallel.py runs it at world 8")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the exchange path on a single-rank RCCL group when launched without torch.distributed.run")
    ap.add_argument("--sync-exchange", action="store_true",
                    help="N>1: plain form of the default schedule -- start the collectives behind the backward and wait for all of "
                         "them before anything else (no overlap of the SH rebuild with the dense sum)")
    ap.add_argument("--stale-overlap", action="store_true",
                    help="N>1: round 2's default -- the collectives of step k are waited for in step k+2 and overlap the next "
                         "view's render (two gradient buffers).  Only valid when the optimizer may apply one-step-stale "
                         "gradients: a loop that updates the parameters between views needs the exchange finished first, "
                         "which is what the default (in-step) schedule does")
    ap.add_argument("--densify-stats", action="store_true",
                    help="N>1: also all-reduce the densification statistics of vanilla 3DGS every step (radii MAX, "
                         "viewspace-gradient norm and visibility count SUM; gaussian_model.py:404-407)")
    return ap.parse_args()


def self_launch_if_needed(args):
    """`python bench.py --gpus N` from a bare shell: become N ranks (one per GPU) under torch.distributed.run."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def dry_run_plumbing(args, world, rank):
    """bench.py's multi-rank path without a GPU (see --dry-run-plumbing): toy packets in the real layout through the real
    SlotSumExchange on gloo.  Every Gaussian has a row in every view; row of view v = v + 1 in all twelve floats; the toy combine
    pass adds the views' rows in view order into the opacity gradient, so every rank must end with world (world + 1) / 2."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from frosting_amd.parallel import SUM_HDR_WORDS, SUM_ROW_FLOATS, SlotSumExchange, sum_packet_words
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = args.points or 20000
    shapes = {"means3D": (P, 3), "scales": (P, 3), "rotations": (P, 4), "opacities": (P, 1), "shs": (P, 16, 3)}

    def layout(n):
        nblk = (n + 63) // 64
        b_at = SUM_HDR_WORDS + 2 * nblk
        return nblk, b_at, (b_at + nblk + 3) // 4 * 4

    def packer(ex, c, dest):
        first, n = ex.chunks[c]
        cap = ex.capacity[c]
        nblk, b_at, r_at = layout(n)
        w = np.zeros(sum_packet_words(n, cap), dtype=np.int32)
        w[0:6] = [min(n, cap), n, n, cap, first, 0x46534d36]
        bits = np.zeros(nblk * 64, dtype=bool)
        bits[:n] = True
        w[SUM_HDR_WORDS:b_at] = np.packbits(bits.reshape(nblk, 64), axis=1, bitorder="little").view(np.int32).reshape(-1)
        w[b_at:b_at + nblk] = 64 * np.arange(nblk, dtype=np.int32)
        w[r_at:r_at + SUM_ROW_FLOATS * min(n, cap)] = np.full(SUM_ROW_FLOATS * min(n, cap), rank + 1, dtype=np.float32).view(np.int32)
        dest.copy_(torch.from_numpy(w))

    def combiner(ex, c, packets, n_views, seq):
        first, n = ex.chunks[c]
        _, _, r_at = layout(n)
        acc = torch.zeros(n)
        wants, over = [], False
        for v in range(n_views):
            w = packets[v].numpy()
            assert int(w[2]) == n and int(w[4]) == first and int(w[5]) == 0x46534d36
            wants.append(int(w[1]))
            over |= int(w[1]) > int(w[3])
            k = min(n, int(w[3]))
            acc[:k] += torch.from_numpy(w[r_at:r_at + SUM_ROW_FLOATS * k].view(np.float32).reshape(k, SUM_ROW_FLOATS)[:, 0].copy())
        ex.views["opacities"][first:first + n, 0] = acc
        st = ex.status[c]
        st[0] = (seq << 32) | int(over)
        st[1:1 + n_views] = torch.tensor([(seq << 32) | x for x in wants], dtype=torch.int64)

    ex = SlotSumExchange(shapes, "cpu", dist.group.WORLD, chunks=args.chunks, packer=packer, combiner=combiner)
    ex.set_params({})

    def step():
        ex.note_view(dry=True)
        ex.start()
        ex.finish_in_step()

    for _ in range(args.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    good = torch.tensor([1.0 if bool((ex.views["opacities"] == world * (world + 1) / 2).all()) else 0.0])
    dist.all_reduce(good, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"metric": "train views/sec (fwd+bwd raster)", "value": world * args.steps / float(dt), "unit": "views/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * float(dt) / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "dry_run": True, "dry_run_note": "launch path and collectives only, toy packets on the CPU: NOT a measurement",
                          "all_ranks_agree": bool(good.item() == 1.0),
                          "config": {"workload": "dry run of the multi-rank plumbing", "P": P, "ranks": world, "backend": "gloo",
                                     "exchange": "slotsum", "chunks": len(ex.chunks), "capacity_rows": list(ex.capacity),
                                     "repacks": ex.stats["repacks"], "parallelism": f"view-parallel x{world}"}}))
    dist.destroy_process_group()


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
    """HBM bytes per launch of `stage` from the committed rocprofv3 PMC passes of this very workload
    (profiles/r0N_pmc_traffic.json, newest round first; counters cannot be read from inside the process).
    None when the run is not the profiled configuration; the string "stale" when the newest committed passes were taken
    with other kernels than the ones running now (their file carries the sha256 of the kernel sources it was measured
    with, tools/collect_traffic.py; a file without one counts as stale)."""
    from frosting_amd import _lib, scenes
    if cfg_name != "c3" or P != scenes.CONFIGS["c3"]["P"]:
        return None
    mine = _lib.build_fingerprint()["kernel_sources_sha256"]
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
        try:
            doc = json.load(open(path))
        except Exception:
            continue
        if doc.get("build", {}).get("kernel_sources_sha256") != mine:
            return "stale"
        per = doc["per_launch"]
        if stage == "*":      # the whole op: every stage of one view
            return {"bytes": sum(v["hbm_bytes_corrected"] for v in per.values()), "source": f"profiles/{rnd}_pmc_traffic.json"}
        return per[stage]["hbm_bytes_corrected"]
    return None


# what one wave-instruction costs a SIMD: tools/micro/pk_rate.hip on this GPU (2048 workgroups x 256 threads of independent
# chains): a plain f32 VALU op 2.8 cycles at the 2.4 GHz the tool assumes (v_exp / v_rcp ~10, v_add_f32_dpp ~7 -- a kernel's
# mix costs more than this floor)
VALU_CYCLES_PER_WAVE_INSTR = 2.8
SIMDS = 256 * 4
CLOCK_HZ = 2.4e9
STAGE_KERNELS = {"preprocess": ("preprocess_fwd_kernel",), "scan": ("colsum_kernel", "reorder_kernel"), "scatter": ("scatter_rows_kernel",),
                 "sort": ("sort_tiles_lds_kernel",), "blend_fwd": ("blend_fwd_kernel",), "blend_bwd": ("blend_bwd_kernel",),
                 "preprocess_bwd": ("preprocess_bwd_kernel",)}


def pmc_valu(stage: str, cfg_name: str, P: int, launch_ms: float):
    """The compute side of the dominant kernel's roofline: vector wave-instructions per launch from the committed SQ counter
    passes (profiles/r0N_pmc_sq.json, same staleness rule as pmc_traffic) -> the time the chip's 1024 SIMDs need to ISSUE them
    at the measured cost of a plain f32 VALU op, and which share of the kernel's launch that is.  A kernel near 1 is bound by
    instruction issue, whatever its HBM fraction says."""
    from frosting_amd import _lib, scenes
    if cfg_name != "c3" or P != scenes.CONFIGS["c3"]["P"] or stage not in STAGE_KERNELS:
        return None
    mine = _lib.build_fingerprint()["kernel_sources_sha256"]
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_sq.json")
        try:
            doc = json.load(open(path))
        except Exception:
            continue
        if doc.get("build", {}).get("kernel_sources_sha256") != mine:
            return "stale"
        ks = [v for k, v in doc["kernels"].items() if any(n in k for n in STAGE_KERNELS[stage])]
        if not ks:
            return None
        insts = sum(v.get("SQ_INSTS_VALU", 0) for v in ks)
        floor_ms = 1e3 * insts * VALU_CYCLES_PER_WAVE_INSTR / SIMDS / CLOCK_HZ
        return {"wave_instr": insts, "cycles_per_instr": VALU_CYCLES_PER_WAVE_INSTR, "simds": SIMDS, "clock_GHz": CLOCK_HZ / 1e9,
                "issue_floor_ms": floor_ms, "issue_frac": floor_ms / launch_ms if launch_ms else None,
                "lds_bank_conflict_cycles": sum(v.get("SQ_LDS_BANK_CONFLICT", 0) for v in ks),
                "source": f"profiles/{rnd}_pmc_sq.json (SQ_INSTS_VALU per launch) x tools/micro/pk_rate.hip (cycles per plain f32 wave-instruction)"}
    return None


def cpu_baseline(cfg_name: str, P: int, backward: bool = True, torch_budget_s: float = 20.0):
    """Two CPU baselines on the host cores for ONE view of the same workload -- reported, not a target:
      * `torch` (the headline entry, what BASELINE.json's north_star names): the alpha blend -- forward, and with
        `backward` its autograd gradients -- in PURE PyTorch on the CPU (oracle/torch_blend.py), on a bounded sample of
        the view's tiles (every k-th tile, k chosen so that the sample takes about `torch_budget_s` seconds), scaled to
        the whole image.  It times the blend stage ALONE from the C port's 2-D state (no projection, no SH, no binning,
        no sort): an upper bound on what a pure-PyTorch rasterizer would reach;
      * `port`: the whole op (preprocess -> sort -> blend -> backward) as the OpenMP C restatement of the reference
        algorithm (oracle/gs_oracle.c), one full view."""
    import numpy as np
    import torch
    from frosting_amd import scenes
    from oracle import gs_oracle as G
    from oracle import torch_blend as TB
    G.build()
    scene, cam, bg = scenes.config_scene(cfg_name, 0, P=P)
    kw = dict(means3D=scene.means3D.numpy(), opacities=scene.opacities.numpy(), viewmatrix=cam.viewmatrix.numpy(),
              projmatrix=cam.projmatrix.numpy(), campos=cam.campos.numpy(), bg=bg.numpy(), width=cam.image_width,
              height=cam.image_height, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, shs=scene.shs.numpy(),
              scales=scene.scales.numpy(), rotations=scene.rotations.numpy(), sh_degree=scene.sh_degree)
    t0 = time.perf_counter()
    st = G.forward(**kw)
    gpix = (np.sign(st["out_color"] - 0.5) / st["out_color"].size).astype(np.float32)
    if backward:
        G.backward(st, gpix)
    dt = time.perf_counter() - t0
    what = "fwd+bwd" if backward else "forward"
    port = {"value": 1.0 / dt, "unit": "views/s", "cores": G.num_threads(), "kind": "port",
            "sample": f"1 view {what} of {cfg_name} at P={P} (R={st['num_rendered']}), {dt:.2f} s "
                      f"wall, OpenMP C port of the reference algorithm (oracle/gs_oracle.c), whole op"}
    try:
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        W, H = cam.image_width, cam.image_height
        T = st["ranges"].shape[0]
        args_t = (t(st["means2D"]), t(st["conic_opacity"]), t(st["rgb"]), t(st["ranges"].astype(np.int64)),
                  t(st["point_list"].astype(np.int64)), bg, W, H)
        g_t = t(gpix) if backward else None
        probe = list(range(T // 2, T, max(1, T // 16)))[:8]           # a few tiles from the middle rows: the rate
        r = TB.render(*args_t, tiles=probe, dL_dimage=g_t)
        per_tile = r["seconds"] / len(probe)
        stride = max(1, int(np.ceil(per_tile * T / torch_budget_s)))
        tiles = list(range(0, T, stride))
        r = TB.render(*args_t, tiles=tiles, dL_dimage=g_t)
        frac = float(sum(int(st["ranges"][i, 1]) - int(st["ranges"][i, 0]) for i in tiles)) / max(1, st["num_rendered"])
        est = r["seconds"] / max(frac, 1e-9)                          # scaled by the share of the list entries sampled
        out = {"value": 1.0 / est, "unit": "views/s", "cores": torch.get_num_threads(), "kind": "torch",
               "sample": f"alpha blend {what} ALONE (no projection / SH / binning / sort) of 1 view of {cfg_name} at P={P}: every "
                         f"{stride}-th of the {T} tiles ({len(tiles)} tiles, {100 * frac:.1f} % of the {st['num_rendered']} list entries) "
                         f"in {r['seconds']:.1f} s, scaled to the whole image; pure PyTorch float32 on the CPU with autograd "
                         f"(oracle/torch_blend.py), {torch.get_num_threads()} torch threads",
               "port": port}
        return out
    except Exception as ex:       # the C port alone is still a baseline
        port["torch_error"] = repr(ex)
        return port


def reference_on_this_gpu(scene_d, cam_d, bg_d, gpix, backward: bool, iters: int = 20, warm: int = 5, keep=None):
    """The reference's own rasterizer (oracle/_ref fast build = hipcc defaults, compiled from /root/reference by
    oracle/build_ref.sh) on the same tensors: 'the reference on MI355X' beside ours.  Baseline leg only.
    SURVEY 8(d)'s protocol: `warm` untimed + `iters` timed iterations, torch.cuda.Event around the forward and around the
    backward (the reference launches on the legacy default stream, which is torch's current stream here), median.
    keep (bool [P], C4): the reference has no skip flag -- Frosting compacts its five parameter tensors with the occlusion
    mask in front of every render (frosting_model.py:1564-1586); that compaction is timed with the forward."""
    import torch
    from oracle import ref_rasterizer as REF
    if not REF.available("fast"):
        return None
    cam = dict(viewmatrix=cam_d.viewmatrix, projmatrix=cam_d.projmatrix, campos=cam_d.campos, bg=bg_d, width=cam_d.image_width,
               height=cam_d.image_height, tanfovx=cam_d.tanfovx, tanfovy=cam_d.tanfovy, sh_degree=scene_d.sh_degree, variant="fast")

    def inputs():
        t = dict(means3D=scene_d.means3D, opacities=scene_d.opacities, shs=scene_d.shs, scales=scene_d.scales, rotations=scene_d.rotations)
        return t if keep is None else {k: v[keep] for k, v in t.items()}

    fw, bw = [], []
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for it in range(warm + iters):
        ev[0].record()
        R, _, _, st = REF.forward(**inputs(), **cam)
        ev[1].record()
        if backward:
            REF.backward(st, gpix)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= warm:
            fw.append(ev[0].elapsed_time(ev[1]))
            bw.append(ev[1].elapsed_time(ev[2]))
    return {"forward_ms": statistics.median(fw), "backward_ms": statistics.median(bw) if backward else None,
            "ms_per_step": statistics.median(fw) + (statistics.median(bw) if backward else 0.0), "num_rendered": int(R),
            "note": "the reference's own CUDA sources built for gfx950 with hipcc (oracle/_ref, fast variant): "
                    f"{warm} warm-up + {iters} timed iterations, torch.cuda.Event around forward and backward, median "
                    "(incl. its allocations and the zero-fill of its gradient outputs" +
                    ("; forward incl. the five boolean compactions by the occlusion mask" if keep is not None else "") + ")"}


def side_config(name, dev, torch, scenes, M, ViewParallelRasterizer, _lib, steps: int = 20, with_reference: bool = False):
    """One of the other BASELINE configs on this GPU: ms per step over `steps` steps (wall clock between two
    synchronisations) and the fraction of the 8 TB/s roofline of ITS algorithmic bytes (C4 adds the
    triangle raster's 12 bytes per vertex of every face and 16 bytes per pixel, and the 9 bytes per Gaussian of the
    mask gather)."""
    cfg = scenes.CONFIGS[name]
    shell = None
    if cfg.get("kind") == "shell":
        shell, cam, bg = scenes.config_shell_scene(name, 0)
        scene = shell.scene
    else:
        scene, cam, bg = scenes.config_scene(name, 0)
    backward = name != "c2"
    vp = ViewParallelRasterizer(scene.to(dev), dev)
    cam_d, bg_d = cam.to(dev), bg.to(dev)
    ctx = M.RasterizeGLContext() if shell is not None else None
    if shell is not None:
        verts_d, faces_d, cell_d = shell.verts.to(dev), shell.faces.to(dev), shell.cell.to(dev)

    def mask():
        if shell is None:
            return None
        return M.occlusion_keep_mask(verts_d, faces_d, cam_d.projmatrix, cam.image_height, cam.image_width, cell_d, 0, ctx)

    img, radii = vp.forward(cam_d, bg_d, keep_mask=mask())
    g, _ = scenes.l1_target_grad(img.cpu(), 11)
    g = g.to(dev)

    def step():
        vp.forward(cam_d, bg_d, keep_mask=mask(), forward_only=not backward)    # (C2 is a forward-only config: frg_forward_args::forward_only)
        if backward:
            vp.backward(g, 0)

    def timed(fn, n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return 1e3 * (time.perf_counter() - t0) / n
    # warm-up by TIME, not by count: the pass follows host-side scene building, and 30 steps of a 0.1 ms forward (C2)
    # are over before the GPU has left its idle clocks -- such a run reported 3.7 ... 4.5 ms per step
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.4:
        for _ in range(30):
            step()
        torch.cuda.synchronize(dev)
    if name == "c2":
        steps = max(steps, 200)          # a 0.1 ms forward: 20 steps would be a 2 ms measurement
    ms = timed(step, steps)
    P, V, R = scene.P, int((radii > 0).sum()), int(vp.true_num_rendered)
    N = cam.image_width * cam.image_height
    T = ((cam.image_width + 15) // 16) * ((cam.image_height + 15) // 16)
    B = (312 * P + 614 * V + 160 * R + 40 * N + 8 * T) if backward else (28 * P + 311 * V + 84 * R + 20 * N + 8 * T)
    out = {"workload": f"{name}: {P} Gaussians, SH deg 3, {cam.image_width}x{cam.image_height}, " +
                       ("forward only" if not backward else "mesh occlusion raster + cull + forward + backward" if shell else "forward+backward"),
           "steps": steps, "ms_per_step": ms, "value": 1e3 / ms, "unit": "views/s", "visible": V, "num_rendered": R}
    if shell is not None:
        F = int(shell.faces.shape[0])
        B += 12 * 3 * F + 16 * N + 9 * P
        out["mesh_triangles"] = F
        out["mesh_raster_ms"] = timed(mask, steps)
        out["mesh_raster_note"] = "triangle raster + visible-face mask + per-Gaussian keep flag (the cull_mask() part of the step), timed on its own"
    out["algorithmic_bytes_per_view"] = B
    out["frac"] = B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    if with_reference:      # ours beside the reference's own code on this GPU, for every config
        try:
            keep = mask()
            ref = reference_on_this_gpu(scene.to(dev), cam_d, bg_d, g, backward, keep=None if keep is None else keep.bool())
            if ref:
                out["reference_on_mi355x"] = ref
                out["speedup_vs_reference"] = ref["ms_per_step"] / ms
        except Exception as ex:     # informative: never lose our number over it
            out["reference_on_mi355x"] = {"error": repr(ex)}
    return out


def main():
    args = parse_args()
    self_launch_if_needed(args)

    import torch
    from frosting_amd import _lib, scenes
    from frosting_amd import mesh as M
    from frosting_amd.parallel import DensificationStats, ViewParallelRasterizer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FRG_BENCH_ONE_GPU"):   # functional test: every rank on GPU 0
        local_rank = 0
    if args.dry_run_plumbing:
        if args.backend != "gloo":
            raise SystemExit("--dry-run-plumbing runs on the CPU: add --backend gloo")
        return dry_run_plumbing(args, world, rank)
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
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}: reporting n_gpus={world}", file=sys.stderr)

    extras = not (args.no_extras or args.no_tight_pass)
    cfg = scenes.CONFIGS[args.config]
    P = args.points or cfg["P"]
    shell = None
    n_views = max(1, min(8, args.views))
    view_ids = [(rank + k) % 8 for k in range(n_views)]       # this rank's cameras, in the order it renders them
    if cfg.get("kind") == "shell":
        shell, cam, bg = scenes.config_shell_scene(args.config, view_ids[0], P=P)
        scene = shell.scene
    else:
        scene, cam, bg = scenes.config_scene(args.config, view_ids[0], P=P)
    cams = [cam] + [scenes.ring_camera(v, cfg["width"], cfg["height"], cfg["fx"], cfg["fy"]) for v in view_ids[1:]]
    do_backward = args.config != "c2"
    for kv in args.option:
        k, v = kv.split("=")
        if _lib.set_option(k, int(v)) < 0 and "unknown option" in _lib.last_error():
            raise SystemExit(f"--option {kv}: {_lib.last_error()}")
    _lib.set_option("exact_blend", 1 if args.exact else 0)
    _lib.set_option("profile", 0 if args.no_stage_timers else 1)
    _lib.set_option("tight_binning", 1 if args.tight_binning else 0)
    scene_d = scene.to(dev)
    reduce_probe = None
    auto_exchange = args.exchange == "auto"
    if auto_exchange:
        args.exchange = "factored"
    exchange_probe = None
    auto_reduce = args.reduce == "auto"
    if auto_reduce:
        args.reduce = "allreduce"
    vpr = ViewParallelRasterizer(scene_d, dev, process_group=dist.group.WORLD if dist else None,
                                 factor_sh=(args.exchange in ("factored", "sparse")), deferred_counters=args.deferred_counters,
                                 reduce=args.reduce, sparse=(args.exchange == "sparse"), slotsum=(args.exchange == "slotsum"),
                                 chunks=args.chunks, phase1_in_pieces=args.phase1_in_pieces)
    if auto_reduce and dist is not None and world > 1 and args.backend == "nccl" and args.exchange in ("factored", "allreduce"):
        # the sum of the dense part timed both ways on this run's buffers, max over ranks; the faster plan is used
        from frosting_amd.parallel import probe_reduce_plan
        reduce_probe, args.reduce = probe_reduce_plan(vpr.exchanges)
        for ex in vpr.exchanges:
            ex.flat.zero_()
    exchanging = dist is not None and do_backward
    cams_d = [c.to(dev) for c in cams]
    cam_d = cams_d[0]
    bg_d = bg.to(dev)
    mesh_ctx = M.RasterizeGLContext() if shell is not None else None
    if shell is not None:
        verts_d, faces_d, cell_d = shell.verts.to(dev), shell.faces.to(dev), shell.cell.to(dev)

    def cull_mask(c_d=None):
        """C4: triangle occlusion raster of the shell's base mesh -> visible faces -> per-Gaussian keep flag
        (frosting_model.py:1524-1539,1564-1586), every step (the per-frame inference form)."""
        if shell is None:
            return None
        c_d = c_d or cam_d
        if args.composed_mask:
            fm = M.visible_face_mask(verts_d, faces_d, c_d.projmatrix, cam.image_height, cam.image_width, mesh_ctx)
            return M.occlusion_mask_from_face_mask(cell_d, fm)
        return M.occlusion_keep_mask(verts_d, faces_d, c_d.projmatrix, cam.image_height, cam.image_width, cell_d, 0, mesh_ctx)

    # one pass over this rank's cameras: the fixed loss gradient dL/dimage of every view, and what each view holds
    gpixs, view_V, view_R = [], [], []
    for k, c_d in enumerate(cams_d):
        image, radii_k = vpr.forward(c_d, bg_d, keep_mask=cull_mask(c_d))
        vpr.finish()
        g_k, _ = scenes.l1_target_grad(image.cpu(), 20241022 + view_ids[k])
        gpixs.append(g_k.to(dev))
        view_V.append(int((radii_k > 0).sum()))
        view_R.append(int(vpr.true_num_rendered))
    image, radii = vpr.forward(cam_d, bg_d, keep_mask=cull_mask())
    vpr.finish()
    gpix = gpixs[0]

    # Exchange schedule (N > 1).  Default, IN-STEP: the collectives are enqueued right behind the backward and are all
    # complete when the step ends -- what a training loop needs that updates the parameters before it renders the next
    # view (every loop of the reference: refine.py:464-571, gaussian_splatting/train.py:79-130).  Inside the step the
    # all-gather of the colour gradients is waited for alone and the SH rebuild it feeds runs beside the sum of the dense
    # part.  --stale-overlap: round 2's software pipeline (the collectives of step k are waited for in step k+2 and
    # overlap the next render; two buffers) -- valid only for one-step-stale gradients.
    counter = [0]
    cycle = [True]            # False: the fixed-camera loop of rounds 1-3 (timed after the timed region, field 'fixed_view')
    exchange_on = [exchanging]
    schedule = ["stale" if args.stale_overlap else "sync" if args.sync_exchange else "in-step"]
    dstats = DensificationStats(P, dev, dist.group.WORLD) if (dist is not None and args.densify_stats) else None

    def step():
        ex = exchange_on[0]
        stale = ex and schedule[0] == "stale"
        slot = counter[0] % 2 if stale else 0
        k = counter[0] % n_views if cycle[0] else 0      # this step's camera
        c_d, g_d = cams_d[k], gpixs[k]
        counter[0] += 1
        vpr.forward(c_d, bg_d, keep_mask=cull_mask(c_d), forward_only=not do_backward and not args.keep_for_backward)
        if not do_backward:
            vpr.finish()
            return
        if stale:
            # the exchange launched two steps ago on this buffer: its collectives are waited for here, its
            # SH rebuild runs on a side stream under the backward below
            vpr.prefetch_exchange(slot)
        overlapped = ex and schedule[0] == "in-step" and not args.deferred_counters
        if overlapped:
            # in-step, factored plan: the all-gather of the colour-gradient payloads leaves after phase 1 of the backward
            # (blend + slot sums) and travels while phase 2 (the per-Gaussian chain) computes; slot-sum plan: phase 1 only,
            # its sums are packed and travel, the chain runs in the exchange's combine pass
            vpr.backward_overlapped(g_d, slot)
        else:
            vpr.backward(g_d, slot, local=not ex)          # writes straight into the flat gradient buffer
        if not vpr.finish():                 # deferred counters: more instances than the arena holds -> redo
            vpr.forward(c_d, bg_d, deferred=False, keep_mask=cull_mask(c_d))
            vpr.backward(g_d, slot)
        if stale:
            vpr.wait_exchange(slot)          # join the rebuild before this buffer's collectives start again
            vpr.start_exchange(slot)
        elif ex and schedule[0] == "sync":
            vpr.start_exchange(slot)
            vpr.wait_exchange(slot)
        elif ex:
            vpr.exchange_in_step(slot, started=overlapped)
        if ex and dstats is not None:
            dstats.update(vpr.radii, vpr.dL_dmeans2D)

    def drain():
        if exchanging:
            vpr.wait_exchange(0)
            vpr.wait_exchange(1)

    def timed(nsteps, fn=step, with_drain=True):
        """(wall seconds, per-step GPU milliseconds from one event per step boundary on the launch stream)"""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nsteps + 1)]
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(nsteps):
            fn()
            evs[i + 1].record()
        if with_drain:
            drain()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        return dt, [evs[i].elapsed_time(evs[i + 1]) for i in range(nsteps)]

    timers = not args.no_stage_timers
    # N > 1: the first steps of a plan are a TRIAL -- the slot-sum path's multi-rank collectives have only ever run on one GPU
    # (single-rank RCCL, two ranks sharing a GPU over gloo).  A rank on which they raise tells the others (one MAX all-reduce),
    # and every rank falls back to the next plan down: slotsum -> factored -> allreduce.  (A collective that HANGS cannot be
    # caught here.)  The plan that ran is in config.exchange, a fallback in config.exchange_fallback.
    exchange_fallback = None
    if exchanging and world > 1:
        for nxt in ("factored", "allreduce", None):
            failed, err = 0.0, None
            try:
                for _ in range(2):
                    step()
                drain()
            except Exception as ex:      # noqa: BLE001
                failed, err = 1.0, repr(ex)
            try:
                flag = torch.tensor([failed], dtype=torch.float64, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                failed = float(flag.item())
            except Exception as ex:      # noqa: BLE001
                raise SystemExit(f"bench.py: the process group itself fails: {ex!r}")
            if not failed:
                break
            if nxt is None or args.exchange == "allreduce":
                raise SystemExit(f"bench.py: every exchange plan failed; last error: {err}")
            exchange_fallback = {"from": args.exchange, "to": nxt if args.exchange != nxt else "allreduce", "error": err}
            args.exchange = exchange_fallback["to"]
            torch.cuda.synchronize(dev)
            vpr.set_exchange_plan(args.exchange, reduce=args.reduce)
    for _ in range(max(0, args.spinup_steps)):
        step()
    drain()
    if auto_exchange and exchanging and world > 1 and (args.backend == "nccl" or args.probe_exchange) and schedule[0] != "stale":
        # both row-level plans timed on this run's scene, max over ranks; the faster one runs (the same on every rank: the
        # choice is made from the reduced times).  Anything going wrong in here must not cost the run: the factored plan,
        # measured since round 2, is the fallback.
        try:
            exchange_probe = {}
            for plan in ("slotsum", "factored", "sparse"):
                vpr.set_exchange_plan(plan, reduce=args.reduce)
                timed(3)
                t_, _ = timed(6)
                tt = torch.tensor([t_ / 6], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                exchange_probe[plan] = 1e3 * float(tt.item())
            args.exchange = min(exchange_probe, key=exchange_probe.get)
        except Exception as ex:      # noqa: BLE001
            exchange_probe = {"error": repr(ex)}
            args.exchange = "factored"
        vpr.set_exchange_plan(args.exchange, reduce=args.reduce)
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
    counter[0] = 0                        # the timed steps render views 0, 1, ... of this rank's cycle
    if dist:
        dist.barrier()
    dt, per_step_ms = timed(args.steps)
    if dist:
        dist.barrier()
    timed_views = [view_ids[i % n_views] for i in range(args.steps)]
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
        _lib.set_option("profile", 0)
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- informative passes, all after and outside the timed region -------------------------------------------
    fixed_view = None
    if n_views > 1 and extras:
        # rounds 1-3's loop: one camera rendered over and over (identical tile lists, arena sizes and cache contents
        # every step -- the best case of a trainer; kept for comparison with the earlier rounds' numbers)
        cycle[0] = False
        for _ in range(3):
            step()
        drain()
        dt_f, ms_f = timed(args.steps)
        if dist:
            tf = torch.tensor([dt_f], dtype=torch.float64, device=dev)
            dist.all_reduce(tf, op=dist.ReduceOp.MAX)
            dt_f = float(tf.item())
        fixed_view = {"view": view_ids[0], "ms_per_step": 1e3 * dt_f / args.steps, "median_ms_per_step": statistics.median(ms_f),
                      "value": world * args.steps / dt_f, "unit": "views/s", "num_rendered": view_R[0],
                      "note": "the same steps on ONE fixed camera (the timed loop of rounds 1-3), after the timed region"}
        cycle[0] = True
    cycle[0] = False        # the passes below compare modes on one camera
    compute_only, other_schedule = None, None
    if exchanging:
        def max_over_ranks(x):
            t_ = torch.tensor([x], dtype=torch.float64, device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            return float(t_.item())
        # the same steps without the exchange: what the collectives add to the step (same camera cycle as the timed region)
        drain()
        cycle[0] = True
        counter[0] = 0
        exchange_on[0] = False
        for _ in range(3):
            step()
        counter[0] = 0
        dt_c, _ = timed(args.steps, with_drain=False)
        exchange_on[0] = True
        compute_only = 1e3 * max_over_ranks(dt_c) / args.steps
        # and the other schedule, for the record (in-step <-> stale overlap; not for the slot-sum plan: its packets can only be
        # packed again -- a view that wants more rows than last step's -- while the backward's workspace is intact, in-step)
        mine = schedule[0]
        if args.exchange != "slotsum":
            schedule[0] = "in-step" if mine == "stale" else "stale"
            for _ in range(4):
                step()
            drain()
            dt_o, _ = timed(args.steps)
            other_schedule = {"schedule": schedule[0], "ms_per_step": 1e3 * max_over_ranks(dt_o) / args.steps}
            other_schedule["exposed_ms_per_step"] = other_schedule["ms_per_step"] - compute_only
            drain()
        schedule[0] = mine
        cycle[0] = False
    single = world == 1 and not exchanging
    tight = None
    if single and extras and not args.tight_binning:
        _lib.set_option("tight_binning", 1)
        for _ in range(3):
            step()
        dt_t, ms_t = timed(args.steps)
        _lib.set_option("tight_binning", 0)
        vpr.forward(cam_d, bg_d, deferred=False, keep_mask=cull_mask())   # leave reference-identical lists in the buffers
        vpr.finish()
        tight = {"ms_per_step": 1e3 * dt_t / args.steps, "median_ms_per_step": statistics.median(ms_t),
                 "value": args.steps / dt_t, "unit": "views/s",
                 "note": "option tight_binning=1 (off in the headline run): tile lists without the instances that provably "
                         "touch no pixel of their tile; image, radii, num_rendered and gradients bit-identical"}
    api_path = None
    if single and extras and shell is None:
        # The path the reference's callers take: diff_gaussian_rasterization.GaussianRasterizer (compiled torch
        # extension _C, autograd, torch's allocator) instead of the persistent arenas of the C-ABI loop above.
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        settings = GaussianRasterizationSettings(
            image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg_d,
            scale_modifier=1.0, viewmatrix=cam_d.viewmatrix, projmatrix=cam_d.projmatrix, sh_degree=scene.sh_degree,
            campos=cam_d.campos, prefiltered=False, debug=False)
        rast = GaussianRasterizer(raster_settings=settings)
        leaves = [t.detach().clone().requires_grad_(do_backward) for t in
                  (scene_d.means3D, scene_d.shs, scene_d.opacities, scene_d.scales, scene_d.rotations)]
        means2D = torch.zeros_like(leaves[0], requires_grad=do_backward)

        def api_step():
            img, _ = rast(means3D=leaves[0], means2D=means2D, shs=leaves[1], colors_precomp=None, opacities=leaves[2],
                          scales=leaves[3], rotations=leaves[4], cov3D_precomp=None)
            if do_backward:
                for t in leaves + [means2D]:
                    t.grad = None                # optimizer.zero_grad(set_to_none=True), as training loops do
                img.backward(gpix)
        for _ in range(25):           # torch's caching allocator needs a few iterations to settle on its block sizes
            api_step()
        dt_a, ms_a = timed(args.steps, fn=api_step, with_drain=False)
        api_path = {"ms_per_step": 1e3 * dt_a / args.steps, "median_ms_per_step": statistics.median(ms_a),
                    "vs_c_abi": (dt_a / args.steps) / (dt / args.steps),
                    "note": "GaussianRasterizer(...)(...) + image.backward(dL) through the compiled extension "
                            "diff_gaussian_rasterization._C (autograd, outputs and scratch from torch's caching allocator)"}
        del leaves, means2D, rast
    ref_gpu = None
    if single and extras and rank == 0 and shell is None and not args.no_cpu_baseline:
        try:
            ref_gpu = reference_on_this_gpu(scene_d, cam_d, bg_d, gpix, do_backward)
        except Exception as ex:
            ref_gpu = {"error": repr(ex)}
    skew = None
    if single and extras and args.config == "c3":
        # a second seeded scene with heavy skew (clusters -> tile lists of 10^4..10^5 entries beside empty tiles,
        # near-camera giants): the >8192 global-memory sort path and the blend makespan under imbalance
        sk = scenes.make_skew_scene(P, cfg["seed"] + 77).to(dev)
        vs = ViewParallelRasterizer(sk, dev)
        img_s, radii_s = vs.forward(cam_d, bg_d)
        g_s, _ = scenes.l1_target_grad(img_s.cpu(), 5)
        g_s = g_s.to(dev)

        def skew_step():
            vs.forward(cam_d, bg_d)
            vs.backward(g_s, 0)
        for _ in range(5):
            skew_step()
        _lib.set_option("profile", 1)
        _lib.stage_times()
        dt_s, ms_s = timed(10, fn=skew_step, with_drain=False)
        st_s = {k: v for k, v in _lib.stage_times().items() if v > 0}
        _lib.set_option("profile", 0)
        from frosting_amd.introspect import State
        rng_s = State(P, cam.image_width, cam.image_height, vs.true_num_rendered, vs.geom.buf, vs.binning.buf, vs.img.buf).ranges
        tl = (rng_s[:, 1] - rng_s[:, 0])
        skew = {"ms_per_step": 1e3 * dt_s / 10, "median_ms_per_step": statistics.median(ms_s), "num_rendered": int(vs.true_num_rendered),
                "visible": int((radii_s > 0).sum()), "tile_list_max": int(tl.max()), "tile_list_mean": float(tl.float().mean()),
                "tiles_over_8192": int((tl > 8192).sum()), "empty_tiles": int((tl == 0).sum()), "stage_ms": st_s,
                "note": "scenes.make_skew_scene: half the Gaussians in four tight clusters, 400 large near-camera Gaussians"}
        del vs, sk

    # BASELINE configs[1] (C2, forward only) and configs[3] (C4: triangle occlusion raster -> face mask -> culled forward
    # + backward), 20 steps each with their own scenes, after and outside the timed region: the driver only runs the
    # default command, so their numbers ride in the headline line
    side = {}
    if single and extras and args.config == "c3" and not args.points:
        for name in ("c2", "c4"):
            try:
                side[name] = side_config(name, dev, torch, scenes, M, ViewParallelRasterizer, _lib, with_reference=True)
            except Exception as ex:
                side[name] = {"error": repr(ex)}

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        views_per_s = world * args.steps / dt
        # per-view figures of the timed steps: the byte model takes their means over the views actually rendered
        idx = [i % n_views for i in range(args.steps)]
        V = sum(view_V[i] for i in idx) / len(idx)
        R = sum(view_R[i] for i in idx) / len(idx)
        N = cam.image_width * cam.image_height
        T = ((cam.image_width + 15) // 16) * ((cam.image_height + 15) // 16)
        from frosting_amd.introspect import State      # report-only: per-tile list statistics
        rng = State(P, cam.image_width, cam.image_height, int(vpr.true_num_rendered), vpr.geom.buf, vpr.binning.buf, vpr.img.buf).ranges
        tile_len = (rng[:, 1] - rng[:, 0]).float()
        B = stage_bytes(P, V, R, N, T)
        if do_backward:
            total_bytes = 312 * P + 614 * V + 160 * R + 40 * N + 8 * T
        else:
            total_bytes = 28 * P + 311 * V + 84 * R + 20 * N + 8 * T
        what = {"c2": "forward only", "c4": "mesh occlusion raster + cull + forward + backward"}.get(args.config, "forward+backward")
        out = {
            "metric": "train views/sec (fwd+bwd raster)" if do_backward else "views/sec (forward raster)",
            "value": views_per_s, "unit": "views/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "median_ms_per_step": statistics.median(per_step_ms),
            "median_note": "median over the timed steps of the GPU time between step boundaries (one event per step on the "
                           "launch stream, rank 0); value and ms_per_step are wall clock over all steps, max over ranks",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {P} Gaussians, SH deg 3, {cam.image_width}x{cam.image_height}, "
                                   f"{what}, 1 view per GPU per step", "P": P,
                       "views": n_views, "views_note": f"step i of rank r renders camera (r + i) % {n_views} of the 8-camera ring; "
                                                       "visible / num_rendered / byte model = means over the timed steps' views",
                       "timed_views_rank0": timed_views,
                       "visible": V, "num_rendered": R, "num_rendered_min": min(view_R), "num_rendered_max": max(view_R),
                       "num_rendered_by_view": dict(zip(map(str, view_ids), view_R)),
                       "instances_per_visible": R / max(V, 1), "tiles": T,
                       "tile_list_mean": float(tile_len.mean()), "tile_list_max": int(tile_len.max()),
                       "parallelism": f"view-parallel x{world}", "ranks": world, "backend": (args.backend if dist else "none"),
                       "exchange": ("none" if not exchanging else
                                    ("slot sums: 48-byte rows {masked dRGB, six pixel moments, three view-direction terms} of the Gaussians with a gradient, index-ordered "
                                     f"behind a bit mask, all-gathered in {len(vpr.exchange.chunks)} fixed-capacity packets per view; every rank "
                                     "runs the per-Gaussian chain for every view's rows in view order in one pass (frg_backward_combine)"
                                     if args.exchange == "slotsum" else
                                     "all 59 floats/Gaussian summed" if args.exchange == "allreduce" else
                                     "sparse: rows (index, 11 floats, dRGB: 64 B) of the Gaussians with a gradient -- counts, "
                                     "one padded all-gather, scatter-add in view order, summed SH gradient rebuilt per rank"
                                     if args.exchange == "sparse" else
                                     "factored: 11 floats/Gaussian summed + all-gather of dRGB (3 floats), "
                                     "summed SH gradient rebuilt per rank") +
                                    ("" if args.exchange in ("sparse", "slotsum") else ", RCCL all-reduce" if args.reduce == "allreduce" else
                                     ", direct: all-to-all of 1/N shards + local sum + all-gather") +
                                    {"in-step": ", in-step schedule: complete before the step ends (valid with a parameter update between "
                                                "views); SH rebuild beside the dense sum",
                                     "sync": ", in-step schedule, no overlap inside the step",
                                     "stale": ", STALE overlap: waited for two steps later, overlapping the next render (2 gradient "
                                              "buffers; one-step-stale gradients only)"}[schedule[0]] +
                                    (", + densification statistics (radii MAX, grad-norm / count SUM)" if dstats is not None else "")),
                       "exchange_bytes_per_rank": (4 * vpr.exchange.wire_floats_per_rank if exchanging else 0),
                       "reduce_probe_ms": reduce_probe, "exchange_probe_ms_per_step": exchange_probe, "exchange_fallback": exchange_fallback,
                       "blend_arithmetic": "exact" if args.exact else "fast", "seed": cfg["seed"],
                       "outputs_written": "all nine gradient tensors of SURVEY 8(d)'s 284 B / Gaussian except dL_dconic (an "
                                          "intermediate the reference's binding never returns, rasterize_points.cu:195): "
                                          "dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations",
                       "binning": "tight" if args.tight_binning else "reference-identical tile lists",
                       "counters": "deferred (no host synchronisation inside the step)" if args.deferred_counters
                                   else "blocking 48-byte read-back per forward"},
            "op_hbm": {"algorithmic_bytes_per_view": total_bytes,
                       "achieved_GBps_per_gpu": total_bytes / (dt / args.steps) / 1e9,
                       "frac_of_8TBps": total_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
        }
        if shell is not None:
            out["config"]["mesh_triangles"] = int(shell.faces.shape[0])
        if do_backward and shell is None:
            # what the view-parallel step is predicted to cost on one 8-GPU node (frosting_amd.parallel.predict_exchange:
            # bytes / nominal xGMI link rate at 80 % efficiency -- arithmetic, not a measurement), for the plan of this run
            # and for the alternatives; render = this run's step without the exchange
            from frosting_amd.parallel import predict_exchange
            render_ms = compute_only if compute_only is not None else ms_per_step
            sched = "in-step" if schedule[0] == "in-step" else "sync"
            rows_frac = (vpr.exchange.sparse_stats["rows_max"] / P) if (exchanging and args.exchange == "sparse" and vpr.exchange.sparse_stats["rows_max"]) else 0.124
            if exchanging and args.exchange == "slotsum" and vpr.exchange.stats["rows_per_view_max"]:
                rows_frac = vpr.exchange.stats["rows_per_view_max"] / P
            plans = [("allreduce", rd, sc) for rd in ("allreduce", "direct") for sc in ("sync",)] + \
                    [("factored", rd, sc) for rd in ("allreduce", "direct") for sc in ("sync", "in-step")] + [("sparse", "allgather", "sync"), ("slotsum", "allgather", "in-step")]
            out["predicted"] = {
                "render_ms": render_ms, "rows_fraction": rows_frac,
                "this_plan": {str(n): predict_exchange(P, 16, n, render_ms, args.exchange, args.reduce, sched, rows_fraction=rows_frac, chunks=args.chunks) for n in (2, 4, 8)},
                "at_8_gpus": {f"{pl}/{rd}/{sc}": round(predict_exchange(P, 16, 8, render_ms, pl, rd, sc, rows_fraction=rows_frac, chunks=args.chunks)["scaling_vs_1gpu"], 2)
                              for pl, rd, sc in plans},
                # the same with every collective priced at ONE bus bandwidth per GPU, whatever its algorithm: the sum of the
                # seven links, what RCCL usually reaches of it, and a pessimistic figure
                "at_8_gpus_by_bus_GBps": {str(bw): {f"{pl}/{sc}": round(predict_exchange(P, 16, 8, render_ms, pl, "direct", sc, bus_GBps=bw, rows_fraction=rows_frac, chunks=args.chunks)["scaling_vs_1gpu"], 2)
                                                    for pl, sc in (("allreduce", "sync"), ("factored", "in-step"), ("sparse", "sync"), ("slotsum", "in-step"))}
                                          for bw in (1071, 450, 300)},
                "note": "scaling = N x render / (render + exposed exchange); link arithmetic: RCCL's all-reduce priced as a ring "
                        "bound by one xGMI link (pessimistic), 'direct' = reduce-scatter + all-gather over all seven links; bus "
                        "arithmetic: incoming bytes / bus bandwidth; slotsum: the packets' all-gather pipelined with the combine pass over "
                        "--chunks ranges, local terms (pack, combine pass, the chain phase 1 leaves out) measured on one MI355X "
                        "(frosting_amd.parallel.SLOTSUM_LOCAL_MS); see DESIGN.md section 5"}
        if compute_only is not None:
            out["exchange_timing"] = {"schedule": schedule[0], "ms_per_step_without_exchange": compute_only,
                                      "exposed_ms_per_step": ms_per_step - compute_only, "other_schedule": other_schedule,
                                      "note": "the same steps with the collectives switched off, and with the other exchange "
                                              "schedule, both timed after the timed region"}
        if dom_ms and dom_ms > 0:
            dom = dom_stage
            ach = B[dom] / (dom_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, args.config, P),
                               "algorithmic_bytes_per_launch": B[dom], "avg_launch_ms": dom_ms,
                               "timed": "hipEvents around this kernel in every timed step",
                               "valu": pmc_valu(dom, args.config, P, dom_ms)}
            out["roofline"]["build"] = _lib.build_fingerprint()
            out["stage_ms"] = stage_avg
            out["stage_ms_note"] = "all stages, 5 extra steps after the timed region"
        if fixed_view:
            out["fixed_view"] = fixed_view
        if tight:
            out["tight_binning"] = tight
        if api_path:
            out["api_path"] = api_path
        if ref_gpu:
            out["reference_on_mi355x"] = ref_gpu
        if skew:
            out["skew_scene"] = skew
        out.update(side)
        whole = pmc_traffic("*", args.config, P)
        if whole == "stale":
            out["op_hbm"]["measured_bytes_per_view"] = "stale"
            out["op_hbm"]["measured_note"] = ("the newest committed PMC passes (profiles/r0N_pmc_traffic.json) were taken with other "
                                              "kernel sources than this build's: not quoted")
        elif whole:
            out["op_hbm"]["measured_bytes_per_view"] = whole["bytes"]
            out["op_hbm"]["measured_over_algorithmic"] = whole["bytes"] / total_bytes
            out["op_hbm"]["measured_note"] = (f"sum over the stages of the HBM-side bytes per launch from the committed rocprofv3 PMC "
                                              f"passes of this workload ({whole['source']}); below 1: L2 / Infinity-Cache reuse")
        if not args.no_cpu_baseline and world == 1 and shell is None:
            try:
                out["cpu_baseline"] = cpu_baseline(args.config, P, do_backward)
            except Exception as ex:  # the baseline is informative; never lose the GPU number over it
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

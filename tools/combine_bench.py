"""The local terms of the slot-sum exchange on ONE GPU: eight views of the C3 scene are rendered one
after the other, phase 1 of each backward leaves its sums, the views' packets are packed where an all-gather would put them,
and the combine pass (frg_backward_combine) runs over all of them -- timed with HIP events:

    one-call backward | phase 1 alone | pack (scan + rows) | combine of N views | the rows each view wanted

and the combined gradient is checked against the accumulation of the eight one-call gradients (bit for bit).
`python tools/combine_bench.py [--config c3] [--views 8] [--chunks 2] [--points N]`; prints one JSON line."""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frosting_amd import scenes                                          # noqa: E402
from frosting_amd.parallel import PARAM_ORDER, ViewParallelRasterizer    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--points", type=int, default=0)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--chunks", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--option", action="append", default=[], help="name=value for frg_set_option")
    ap.add_argument("--side-copy", type=int, default=0,
                    help="workgroups of a copy kernel (tools/micro/side_copy.hip) that streams 7 packets' worth of bytes on a side stream "
                         "WHILE the combine pass runs: a stand-in for incoming all-gather traffic -- reports the pass's time beside it")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    from frosting_amd import _lib
    for kv in a.option:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    cfg = scenes.CONFIGS[a.config]
    P = a.points or cfg["P"]
    scene, _, bg = scenes.config_scene(a.config, 0, P=P)
    vpr = ViewParallelRasterizer(scene.to(dev), dev, slotsum=True, chunks=a.chunks)
    ex = vpr.exchange
    bg_d = bg.to(dev)
    cams = [scenes.ring_camera(v, cfg["width"], cfg["height"], cfg["fx"], cfg["fy"]).to(dev) for v in range(a.views)]
    ev = lambda: torch.cuda.Event(enable_timing=True)
    acc = {n: torch.zeros_like(ex.views[n]) for n in PARAM_ORDER}
    t_one, t_p1, t_pack, gpixs = [], [], [], []
    for rep in range(2):               # the second round is the one that counts (arenas sized, packets at the capacity the first asked for)
        if rep == 1:
            verdicts = ex.combine_local(a.views)
            counts = [sum(c[v] for _, c in verdicts) for v in range(a.views)]
            ex.capacity = [min(n, (int(max(c) * 1.25) // 256 + 1) * 256) for (_, n), (_, c) in zip(ex.chunks, verdicts)]
            ex.packets_all = [None] * len(ex.chunks)
            for n in PARAM_ORDER:
                acc[n].zero_()
            t_one, t_p1, t_pack = [], [], []
        for v, cam in enumerate(cams):
            img, _ = vpr.forward(cam, bg_d)
            if rep == 0:
                g, _ = scenes.l1_target_grad(img.cpu(), 20241022 + v)
                gpixs.append(g.to(dev))
            e = [ev() for _ in range(6)]
            e[0].record(); grads = vpr.backward(gpixs[v], 0); e[1].record()
            for n in PARAM_ORDER:
                acc[n] += grads[n]
            e[2].record(); vpr.backward(gpixs[v], 0, slot_sums=True); e[3].record()
            e[4].record(); ex.pack_local_view(v, a.views); e[5].record()
            torch.cuda.synchronize(dev)
            t_one.append(e[0].elapsed_time(e[1])); t_p1.append(e[2].elapsed_time(e[3])); t_pack.append(e[4].elapsed_time(e[5]))
    for t in ex.views.values():
        t.fill_(float("nan"))
    verdicts = ex.combine_local(a.views)
    torch.cuda.synchronize(dev)
    ok = None
    if not a.no_check:
        ok = all(torch.equal(ex.views[n], acc[n]) for n in PARAM_ORDER) and not any(o for o, _ in verdicts)
    t_cmb = []
    for _ in range(a.iters):
        e0, e1 = ev(), ev()
        e0.record(); ex.combine_local(a.views); e1.record()
        torch.cuda.synchronize(dev)
        t_cmb.append(e0.elapsed_time(e1))
    med = statistics.median
    side = None
    if a.side_copy > 0:
        import ctypes as C
        lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libside_copy.so"))
        lib.side_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        nbytes = 7 * 4 * ex.wire_floats_per_rank // 16 * 16
        src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        s2 = torch.cuda.Stream(dev)
        t_alone, t_cmb2, t_copy = [], [], []
        for rep in range(a.iters):                     # the copy alone (its rate), then the combine pass beside it
            c0, c1 = ev(), ev()
            torch.cuda.synchronize(dev)
            with torch.cuda.stream(s2):
                c0.record(s2); lib.side_copy(src.data_ptr(), dst.data_ptr(), nbytes, a.side_copy, 1, s2.cuda_stream); c1.record(s2)
            torch.cuda.synchronize(dev)
            t_alone.append(c0.elapsed_time(c1))
            e0, e1, c0, c1 = ev(), ev(), ev(), ev()
            with torch.cuda.stream(s2):
                c0.record(s2); lib.side_copy(src.data_ptr(), dst.data_ptr(), nbytes, a.side_copy, 2, s2.cuda_stream); c1.record(s2)
            e0.record(); ex.combine_local(a.views); e1.record()
            torch.cuda.synchronize(dev)
            t_cmb2.append(e0.elapsed_time(e1)); t_copy.append(c0.elapsed_time(c1))
        side = {"workgroups": a.side_copy, "bytes": nbytes, "copy_alone_ms": med(t_alone), "copy_alone_GBps": nbytes / med(t_alone) / 1e6,
                "combine_beside_copy_ms": med(t_cmb2), "copy_of_twice_the_bytes_beside_combine_ms": med(t_copy),
                "combine_slowdown": med(t_cmb2) / med(t_cmb)}
    live = int(((acc["opacities"] != 0).reshape(P, -1).any(1) | (acc["means3D"] != 0).any(1)).sum())
    print(json.dumps({"config": a.config, "P": P, "views": a.views, "chunks": a.chunks, "rows_wanted_per_view": counts,
                      "rows_fraction": max(counts) / P, "gaussians_with_a_row_in_some_view": live, "capacity_rows": sum(ex.capacity),
                      "packet_bytes_per_view": 4 * ex.wire_floats_per_rank, "backward_one_call_ms": med(t_one), "backward_phase1_ms": med(t_p1),
                      "pack_ms": med(t_pack), "combine_ms": med(t_cmb), "combine_ms_all": [round(x, 4) for x in t_cmb],
                      "combine_equals_accumulation_bit_for_bit": ok, "side_copy": side}))


if __name__ == "__main__":
    main()

"""View-parallel path on one GPU: the factored SH exchange (csrc/view_exchange.hip) against the
plain sum of per-view gradients, and the collective plumbing on a single-rank RCCL group."""
import os
import socket

import pytest
import torch

from frosting_amd import scenes
from frosting_amd.parallel import GradientExchange, SlotSumExchange, ViewParallelRasterizer, PARAM_ORDER

pytestmark = pytest.mark.gpu

K_SH0 = 0.28209479177387814


def _per_view(vpr, name, views, dev, P=None):
    """Render `views` one after the other; per-view gradients and exchange payloads."""
    grads, payloads = [], []
    for k in views:
        scene, cam, bg = scenes.config_scene(name, k, P=P)
        img, _ = vpr.forward(cam.to(dev), bg.to(dev))
        gpix, _ = scenes.l1_target_grad(img.cpu(), 555 + k)
        g = vpr.backward(gpix.to(dev), 0, payload=True)
        grads.append({n: g[n].clone() for n in PARAM_ORDER})
        payloads.append(vpr.exchange.own.clone())
    return grads, payloads


@pytest.mark.parametrize("P", [5000, 4807])
def test_rebuilt_sh_gradient_is_the_sum_of_view_gradients_bit_exact(gpu_device, P):
    dev = gpu_device
    scene, _, _ = scenes.config_scene("mini", 0, P=P)
    vpr = ViewParallelRasterizer(scene.to(dev), dev, factor_sh=True)
    views = [0, 3, 5, 6]
    grads, payloads = _per_view(vpr, "mini", views, dev, P=P)
    ex = vpr.exchange
    # the payload's colour gradient is the masked one: DC coefficient = kSH0 * dRGB (backward.cu:36-38)
    for g, pay in zip(grads, payloads):
        drgb = pay[: 3 * P].view(P, 3)
        assert torch.equal(drgb * K_SH0, g["shs"][:, 0, :])
        assert (drgb != 0).any()
    ex.gathered = torch.stack(payloads)
    ex.sh_reducer(ex)                               # HIP: frg_sh_grad_from_views
    want = grads[0]["shs"].clone()
    for g in grads[1:]:
        want = want + g["shs"]                      # view order, like the kernel
    got = ex.views["shs"]
    assert torch.equal(got, want)
    assert float(want.abs().max()) > 0


def test_rebuild_lower_degree_and_zero_views(gpu_device):
    dev = gpu_device
    scene, cam, bg = scenes.config_scene("mini", 0, P=3000)
    scene.sh_degree = 1
    vpr = ViewParallelRasterizer(scene.to(dev), dev, factor_sh=True)
    img, _ = vpr.forward(cam.to(dev), bg.to(dev))
    gpix, _ = scenes.l1_target_grad(img.cpu(), 9)
    g = vpr.backward(gpix.to(dev), 0, payload=True)
    want = g["shs"].clone()
    assert float(want[:, 4:].abs().max()) == 0.0 and float(want[:, 1:4].abs().max()) > 0
    ex = vpr.exchange
    ex.gathered = ex.own.clone()[None]
    ex.views["shs"].fill_(123.0)
    ex.sh_reducer(ex)
    assert torch.equal(ex.views["shs"], want)
    ex.gathered = ex.gathered[:0]                    # no views: all zeros
    ex.sh_reducer(ex)
    assert float(ex.views["shs"].abs().max()) == 0.0


def _slot_sum_views(vpr, name, views, dev, P, world=None, seed0=555, pieces=False):
    """One process playing every rank of the slot-sum exchange: per view the one-call backward (the reference gradient of that
    view) and, from the same forward, phase 1 + the view's packets where an all-gather would have put them."""
    ex = vpr.exchange
    world = world or len(views)
    acc = {n: torch.zeros_like(ex.views[n]) for n in PARAM_ORDER}
    for slot_v, k in enumerate(views):
        scene, cam, bg = scenes.config_scene(name, k, P=P)
        img, _ = vpr.forward(cam.to(dev), bg.to(dev))
        gpix, _ = scenes.l1_target_grad(img.cpu(), seed0 + k)
        gpix = gpix.to(dev)
        g = vpr.backward(gpix, 0)                               # every gradient of this view, one call
        for n in PARAM_ORDER:
            acc[n] += g[n]                                      # the single-process accumulation, in view order
        if pieces:                                              # phase 1 in pieces (frg_backward_args::range_first / range_count): the blend
            for first, n in ex.chunks:                              # backward with the first range, every range's sums by its own call
                vpr.backward(gpix, 0, slot_sums=True, sum_range=(first, n))
        else:
            vpr.backward(gpix, 0, slot_sums=True)               # phase 1 only: the nine sums + their bit mask in the workspace
        ex.pack_local_view(slot_v, world)
    return acc


@pytest.mark.parametrize("name,P,views,chunks,degree,raw,pieces", [("mini", 5000, [0, 3, 5, 6], 1, 3, False, False), ("mini", 4807, [1, 2], 3, 1, False, True),
                                                                  ("c2", 80_000, list(range(8)), 2, 3, False, False), ("c2", 50_000, [0, 1, 2, 4], 2, 3, True, False),
                                                                  ("c2", 80_001, [0, 2, 4, 6, 7], 4, 3, False, True), ("c2", 50_000, [5], 3, 3, True, True)])
def test_slot_sum_combine_is_the_accumulation_of_the_view_gradients_bit_for_bit(gpu_device, name, P, views, chunks, degree, raw, pieces):
    """frg_pack_sum_rows + frg_backward_combine (round 6): the rows of the nine per-Gaussian sums phase 1 leaves, packed per view
    in index order behind a bit mask, and ONE pass that runs the per-Gaussian chain for every view's row in view order --
    against the gradients of the same views from one-call backwards, accumulated in view order in one process: every one of the
    59 floats per Gaussian the same bits; lower SH degree; raw parameters (activation Jacobians per view); several chunks; phase 1
    in one call or in pieces, one per chunk; every tile size of the combine pass (eight views .. a single one)."""
    dev = gpu_device
    scene, _, _ = scenes.config_scene(name, 0, P=P)
    scene.sh_degree = degree
    if raw:
        scene = scenes.Scene(scene.means3D, torch.log(scene.scales), scene.rotations * 1.7, torch.log(scene.opacities / (1 - scene.opacities)),
                             scene.shs, degree)
    vpr = ViewParallelRasterizer(scene.to(dev), dev, slotsum=True, chunks=chunks, raw_params=raw)
    ex = vpr.exchange
    assert isinstance(ex, SlotSumExchange) and len(ex.chunks) == chunks and sum(n for _, n in ex.chunks) == P
    acc = _slot_sum_views(vpr, name, views, dev, P, pieces=pieces)
    for t in ex.views.values():
        t.fill_(float("nan"))                                   # every row must be written by the pass
    verdicts = ex.combine_local(len(views))
    torch.cuda.synchronize(dev)
    total = [0] * len(views)
    for over, counts in verdicts:
        assert not over
        total = [a + b for a, b in zip(total, counts)]
    assert all(0 < t < P for t in total)                        # some Gaussians have a gradient in each view, not all
    for n in PARAM_ORDER:
        assert float(acc[n].abs().max()) > 0
        assert torch.equal(ex.views[n], acc[n]), n


@pytest.mark.parametrize("views,chunks", [([0, 1, 2, 3, 4, 5, 6, 7], 2), ([3], 1)])
def test_slot_sum_combine_with_row_live_writes_only_the_rows_with_a_gradient(gpu_device, views, chunks):
    """frg_combine_args::row_live: the combine pass marks the Gaussians that have a row in some view and leaves the rows of the
    others UNWRITTEN (what frg_backward_args::row_live does for one view); the marked rows are the accumulation's, bit for bit."""
    dev = gpu_device
    P = 150_000
    scene, _, _ = scenes.config_scene("c3", 0, P=P)             # the large image: most Gaussians are reached by no pixel of a view
    vpr = ViewParallelRasterizer(scene.to(dev), dev, slotsum=True, chunks=chunks)
    ex = vpr.exchange
    acc = _slot_sum_views(vpr, "c3", views, dev, P)
    ex.row_live = torch.full((P,), 7, dtype=torch.uint8, device=dev)
    for t in ex.views.values():
        t.fill_(123.0)
    verdicts = ex.combine_local(len(views))
    torch.cuda.synchronize(dev)
    assert not any(o for o, _ in verdicts)
    live = ex.row_live.bool()
    assert set(ex.row_live.unique().tolist()) <= {0, 1} and 0 < int(live.sum()) < P
    has_grad = torch.zeros(P, dtype=torch.bool, device=dev)
    for n in PARAM_ORDER:
        has_grad |= (acc[n].reshape(P, -1) != 0).any(1)
    assert bool((has_grad <= live).all())                       # every Gaussian with a gradient is marked (a marked one may sum to zero)
    for n in PARAM_ORDER:
        got, want = ex.views[n].reshape(P, -1), acc[n].reshape(P, -1)
        assert torch.equal(got[live], want[live]), n
        assert bool((got[~live] == 123.0).all()), n             # untouched
    ex.row_live = None


def test_slot_sum_packets_report_an_overflow_and_fit_after_it(gpu_device):
    """A packet is all-gathered at a fixed capacity; a view that wants more rows says so in its header, the combine pass posts
    the verdict (pinned host memory), and the chunk is packed again with room for it: the sums are still in the workspace."""
    dev = gpu_device
    P, views = 20_000, [0, 1, 2]
    scene, _, _ = scenes.config_scene("mini", 0, P=P)
    vpr = ViewParallelRasterizer(scene.to(dev), dev, slotsum=True, chunks=2)
    ex = vpr.exchange
    acc = _slot_sum_views(vpr, "mini", views, dev, P)
    (o0, c0), (o1, c1) = ex.combine_local(len(views))
    assert not o0 and not o1
    want = [ex.views[n].clone() for n in PARAM_ORDER]
    assert all(torch.equal(w, acc[n]) for w, n in zip(want, PARAM_ORDER))
    ex.capacity = [max(c0) - 5, max(c1) + 3]                     # chunk 0: too small for its fullest view; chunk 1: just enough
    ex.packets_all = [None, None]
    _slot_sum_views(vpr, "mini", views, dev, P)
    (o0, d0), (o1, d1) = ex.combine_local(len(views))
    assert o0 and not o1 and d0 == c0 and d1 == c1               # the wanted counts arrive whatever the capacity
    ex.capacity[0] = max(c0)
    ex.packets_all[0] = None
    _slot_sum_views(vpr, "mini", views, dev, P)
    (o0, _), (o1, _) = ex.combine_local(len(views))
    assert not o0 and not o1
    assert all(torch.equal(ex.views[n], acc[n]) for n in PARAM_ORDER)



def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_exchange_plans_on_single_rank_rccl_group(gpu_device):
    """Both exchange plans through RCCL (world size 1): the collectives run, the pipelined
    start/wait protocol works on HIP streams and a 1-rank sum is the view's own gradient."""
    import torch.distributed as dist
    dev = gpu_device
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        scene, cam, bg = scenes.config_scene("mini", 2, P=6000)
        ref = None
        for factored in (False, True, "sparse", "slotsum"):
            vpr = ViewParallelRasterizer(scene.to(dev), dev, process_group=dist.group.WORLD, factor_sh=factored is True or factored == "sparse",
                                         sparse=factored == "sparse", slotsum=factored == "slotsum")
            img, _ = vpr.forward(cam.to(dev), bg.to(dev))
            gpix, _ = scenes.l1_target_grad(img.cpu(), 77)
            gpix = gpix.to(dev)
            for step in range(4):                    # two buffers, exchange of step k waited at step k+2
                slot = step % 2
                if step % 2:
                    vpr.prefetch_exchange(slot)      # side-stream completion, joined by wait_exchange below
                vpr.forward(cam.to(dev), bg.to(dev))
                vpr.wait_exchange(slot)
                vpr.backward(gpix, slot)
                assert vpr.start_exchange(slot) is not None
            outs = [vpr.wait_exchange(s).clone() for s in (0, 1)]
            torch.cuda.synchronize(dev)
            assert torch.equal(outs[0], outs[1])
            if ref is None:
                ref = outs[0]
            else:
                assert torch.equal(outs[0], ref)     # factored == plain, bit for bit, at one view
            assert float(ref.abs().max()) > 0
            # the in-step schedule with the backward in two calls: the payload's all-gather is enqueued between the phases
            # (backward_overlapped), finish_in_step completes everything -- the same bits again
            vpr.forward(cam.to(dev), bg.to(dev))
            own_before = vpr.exchanges[0].own.clone() if factored in (True, "sparse") else None
            vpr.backward_overlapped(gpix, 0)
            flat = vpr.exchange_in_step(0, started=True).clone()
            torch.cuda.synchronize(dev)
            assert torch.equal(flat, ref)
            if factored in (True, "sparse"):
                assert torch.equal(vpr.exchanges[0].own, own_before)       # the payload phase 1 wrote = the one-call payload
            if factored == "slotsum":    # one view: its rows, fewer than the visible Gaussians; the next step's packets are sized from them
                st = vpr.exchanges[0].stats
                assert 0 < st["rows_wanted_max"] <= int((vpr.radii > 0).sum()) and st["repacks"] == 0
                assert st["rows_wanted_max"] <= sum(vpr.exchanges[0].capacity) <= 6000
            if factored == "sparse":     # rows of the Gaussians with a gradient only: fewer than are visible, packed without a re-pack at this size
                st = vpr.exchanges[0].sparse_stats
                assert 0 < st["rows_own"] <= int((vpr.radii > 0).sum()) and st["rows_max"] == st["rows_own"]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scene_kind", ["ball", "giants"])
def test_backward_in_two_calls_equals_one_call(gpu_device, scene_kind):
    """frg_backward_args::phase: 1 (backward blend + per-Gaussian slot sums, dL_dcolor complete) then 2 (the rest, from
    the sums kept in the workspace) write exactly what the one-call backward writes -- every gradient, bit for bit --
    on the uniform ball and on a frame with near-camera giants (waves of thousands of slots: the 16-wave form runs in
    phase 1, phase 2 needs no second form)."""
    dev = gpu_device
    if scene_kind == "ball":
        scene, cam, bg = scenes.config_scene("c2", 1, P=80_000)
    else:
        scene = scenes.make_skew_scene(20_000, 99, centres=[[0.0, 0.0, 0.0]], cluster_sigma=0.05, cluster_frac=0.2, n_big=120, big_scale=0.3)
        cam, bg = scenes.ring_camera(0, 320, 240, 267.0, 267.0), torch.zeros(3)
    vpr = ViewParallelRasterizer(scene.to(dev), dev)
    cam_d, bg_d = cam.to(dev), bg.to(dev)
    img, _ = vpr.forward(cam_d, bg_d)
    gpix, _ = scenes.l1_target_grad(img.cpu(), 5)
    gpix = gpix.to(dev)
    vpr.backward(gpix, 0)
    want = (vpr.exchange.flat.clone(), vpr.dL_dmeans2D.clone(), vpr.dL_dcolors.clone(), vpr.dL_dcov3D.clone())
    assert float(want[0].abs().max()) > 0
    for t in (vpr.exchange.flat, vpr.dL_dmeans2D, vpr.dL_dcolors, vpr.dL_dcov3D):
        t.fill_(float("nan"))
    vpr.backward(gpix, 0, phase=1)
    assert torch.equal(vpr.dL_dcolors, want[2])                              # complete after phase 1
    assert bool(torch.isnan(vpr.exchange.flat).all())                         # ... and nothing else written yet
    vpr.backward(gpix, 0, phase=2)
    got = (vpr.exchange.flat, vpr.dL_dmeans2D, vpr.dL_dcolors, vpr.dL_dcov3D)
    assert all(torch.equal(a, b) for a, b in zip(want, got))
    # the sums phase 1 left are consumed once, and only by a phase 2 with the same buffers: anything else is refused
    # instead of turning stale workspace contents into gradients
    with pytest.raises(RuntimeError, match="phase 2 without a matching phase 1"):
        vpr.backward(gpix, 0, phase=2)
    vpr.backward(gpix, 0, phase=1)
    vpr.work.buf = torch.empty(vpr.work.buf.numel() + 4096, dtype=torch.uint8, device=dev)    # the arena "grew" between the calls
    with pytest.raises(RuntimeError, match="phase 2 without a matching phase 1"):
        vpr.backward(gpix, 0, phase=2)


def test_deferred_counters_forward_matches_blocking_forward(gpu_device):
    """frg_forward_deferred (no host synchronisation) against frg_forward: same image, radii and
    gradients bit for bit; a too-small capacity is reported by finish() and nothing is rasterized."""
    dev = gpu_device
    scene, cam, bg = scenes.config_scene("c2", 1, P=40000)
    cam_d, bg_d = cam.to(dev), bg.to(dev)
    ref = ViewParallelRasterizer(scene.to(dev), dev)
    img0, radii0 = ref.forward(cam_d, bg_d)
    img0, radii0 = img0.clone(), radii0.clone()
    gpix, _ = scenes.l1_target_grad(img0.cpu(), 3)
    gpix = gpix.to(dev)
    g0 = ref.backward(gpix, 0)
    flat0 = ref.exchange.flat.clone()
    R = ref.num_rendered

    vpr = ViewParallelRasterizer(scene.to(dev), dev, deferred_counters=True)
    vpr.forward(cam_d, bg_d)                       # first view: blocking, sizes the arena
    assert vpr.finish() and vpr.capacity >= R
    for _ in range(3):                             # deferred from here on (sort launches sized from the last view)
        img, radii = vpr.forward(cam_d, bg_d)
        assert vpr.num_rendered == vpr.capacity    # what backward carves its buffers with
        vpr.backward(gpix, 0)
        assert vpr.finish() and vpr.true_num_rendered == R
        assert torch.equal(img, img0) and torch.equal(radii, radii0)
        assert torch.equal(vpr.exchange.flat, flat0)

    # another camera with the sort launches still sized from the previous one
    scene2, cam2, _ = scenes.config_scene("c2", 5, P=40000)
    want, _ = ref.forward(cam2.to(dev), bg_d)
    want = want.clone()
    got, _ = vpr.forward(cam2.to(dev), bg_d)
    assert vpr.finish()
    assert torch.equal(got, want)

    # capacity exceeded: reported, nothing rasterized (background only), then repeated
    vpr.capacity = 1000
    img, _ = vpr.forward(cam_d, bg_d)
    vpr.backward(gpix, 0)
    assert not vpr.finish() and vpr.capacity >= R
    assert torch.equal(img, bg_d[:, None, None].expand_as(img))
    assert float(vpr.exchange.flat.abs().max()) == 0.0
    img, _ = vpr.forward(cam_d, bg_d)
    vpr.backward(gpix, 0)
    assert vpr.finish()
    assert torch.equal(img, img0) and torch.equal(vpr.exchange.flat, flat0)


def _rank_worker(rank, world, port, factored, P, q, sparse=False, slotsum=False):
    """One rank of a 2-rank job on GPU 0 (gloo carries the collectives: RCCL refuses two ranks per device)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        scene, cam, bg = scenes.config_scene("mini", rank + 1, P=P)       # rank k renders view k + 1
        vpr = ViewParallelRasterizer(scene.to(dev), dev, process_group=dist.group.WORLD, factor_sh=factored, sparse=sparse, slotsum=slotsum)
        img, _ = vpr.forward(cam.to(dev), bg.to(dev))
        gpix, _ = scenes.l1_target_grad(img.cpu(), 555 + rank + 1)
        gpix = gpix.to(dev)
        for step in range(3):                                              # pipelined protocol, two buffers
            slot = step % 2
            vpr.forward(cam.to(dev), bg.to(dev))
            vpr.prefetch_exchange(slot)                                    # SH rebuild on a side stream ...
            vpr.backward(gpix, slot)                                       # ... under this backward
            vpr.wait_exchange(slot)                                        # joined before the buffer is exchanged again
            vpr.start_exchange(slot)
        out = []
        for s in (0, 1):                                                   # named views, packed: the flat buffer pads
            vpr.wait_exchange(s)                                           # every segment to a 16-byte boundary
            out.append(torch.cat([vpr.exchanges[s].views[n].reshape(-1) for n in PARAM_ORDER]))
        torch.cuda.synchronize(dev)
        assert torch.equal(out[0], out[1])
        q.put((rank, out[0].cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("factored", [False, True, "sparse", "slotsum"])
def test_two_ranks_sum_equals_single_process_sum(gpu_device, factored):
    import torch.multiprocessing as mp
    world, P = 2, 5000
    sparse, slotsum = factored == "sparse", factored == "slotsum"
    factored = bool(factored) and not slotsum
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, factored, P, q, sparse, slotsum)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: the same two views rendered one after the other, gradients added
    dev = gpu_device
    scene, _, _ = scenes.config_scene("mini", 0, P=P)
    vpr = ViewParallelRasterizer(scene.to(dev), dev, factor_sh=True)
    grads, _ = _per_view(vpr, "mini", [1, 2], dev, P=P)
    want = torch.cat([(grads[0][n] + grads[1][n]).reshape(-1) for n in PARAM_ORDER]).cpu()
    for r in range(world):
        got = torch.from_numpy(res[r])
        if factored or slotsum:   # SH part: in-order sum, bit-exact; dense part: the all-reduce of two terms (slot sums: the in-order sum), exact as well
            assert torch.equal(got, want)
        else:
            torch.testing.assert_close(got, want, rtol=0, atol=0)
    assert (res[0] == res[1]).all()


@pytest.mark.parametrize("exchange", ["slotsum", "factored", "allreduce", "sparse", "auto"])
def test_bench_two_ranks_from_a_bare_shell(gpu_device, exchange):
    """`python bench.py --gpus 2` with no launcher around it: the script re-executes itself under
    torch.distributed.run, both ranks share GPU 0 (FRG_BENCH_ONE_GPU) and exchange over gloo; rank 0 prints
    ONE JSON line that says two ranks ran and what they exchanged."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FRG_BENCH_ONE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--exchange", exchange,
                        "--points", "30000", "--steps", "3", "--warmup", "2", "--spinup-steps", "2", "--no-cpu-baseline",
                        "--no-extras", "--probe-exchange"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["scaling"] == "weak"
    assert out["config"]["exchange_bytes_per_rank"] > 0 and "exchange_timing" in out
    if exchange == "auto":      # both row-level plans were timed and one of them runs
        probe = out["config"]["exchange_probe_ms_per_step"]
        assert set(probe) == {"slotsum", "factored", "sparse"} and all(v > 0 for v in probe.values()), probe
    assert out["value"] > 0 and abs(out["value"] - 2 * 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]


def _sharded_adam_worker(rank, world, port, P, K, steps, q):
    """One rank of a 2-rank job on GPU 0 over gloo: ShardedFlatAdam with the HIP shard update (frg_adam_step_shard)."""
    import torch.distributed as dist
    from frosting_amd.optim import ShardedFlatAdam
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
        lrs = dict(means3D=1.6e-4, scales=5e-3, rotations=1e-3, opacities=5e-2, shs=2.5e-3 / 20)
        opt = ShardedFlatAdam(shapes, lrs, dev, dist.group.WORLD, sh_dc_lr=2.5e-3)
        g0 = torch.Generator().manual_seed(5)
        for k in PARAM_ORDER:
            opt.params[k].copy_(torch.randn(shapes[k], generator=g0))
        for it in range(steps):
            g = torch.Generator().manual_seed(1000 * it + rank)
            opt.step((torch.randn(opt.numel, generator=g) * 0.01).to(dev))
        torch.cuda.synchronize(dev)
        q.put((rank, opt.flat.cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_adam_on_two_ranks_equals_flat_adam_on_the_summed_gradients(gpu_device):
    """SURVEY 8(e), second option, with the HIP kernel: two ranks reduce-scatter their gradients, each updates its half of the
    flat buffer (frg_adam_step_shard: the boundary falls inside the SH rows, off the 48-element DC / rest period, and inside
    a 16-byte group of the unsharded layout's neighbour segment), the halves are all-gathered -- the parameters equal
    FlatAdam.step on the summed gradients, bit for bit, after three steps."""
    import torch.multiprocessing as mp
    from frosting_amd.optim import FlatAdam
    world, P, K, steps = 2, 1031, 16, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_adam_worker, args=(r, world, port, P, K, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    dev = gpu_device
    shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
    lrs = dict(means3D=1.6e-4, scales=5e-3, rotations=1e-3, opacities=5e-2, shs=2.5e-3 / 20)
    ref = FlatAdam(shapes, lrs, dev, sh_dc_lr=2.5e-3)
    g0 = torch.Generator().manual_seed(5)
    for k in PARAM_ORDER:
        ref.params[k].copy_(torch.randn(shapes[k], generator=g0))
    for it in range(steps):
        total = None
        for r in range(world):
            g = torch.randn(ref.numel, generator=torch.Generator().manual_seed(1000 * it + r)) * 0.01
            total = g if total is None else total + g
        ref.step(total.to(dev))
    want = ref.flat.cpu()
    assert float(want.abs().max()) > 0
    for r in range(world):
        assert torch.equal(torch.from_numpy(res[r]), want), r

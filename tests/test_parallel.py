"""View-parallel gradient exchange on CPU: world_size-2 gloo processes."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from frosting_amd.parallel import GradientExchange, PARAM_ORDER


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, P, K, q, reduce="allreduce"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
    ex = GradientExchange(shapes, "cpu", dist.group.WORLD, reduce=reduce)
    g = torch.Generator().manual_seed(100 + rank)
    for k in PARAM_ORDER:  # a rank-specific "per-view gradient"
        ex.views[k].copy_(torch.randn(shapes[k], generator=g))
    if rank % 2:
        ex.all_reduce()                      # synchronous form
    else:
        assert ex.start() is not None        # asynchronous form: start, do something else, wait
        _ = torch.ones(10).sum()
        ex.wait()
    assert ex.wait() is ex.flat              # idempotent
    q.put((rank, {k: ex.views[k].numpy().copy() for k in PARAM_ORDER}))  # numpy: plain pickling
    dist.barrier()
    dist.destroy_process_group()


def test_flat_buffer_layout():
    ex = GradientExchange(dict(means3D=(5, 3), scales=(5, 3), rotations=(5, 4), opacities=(5, 1), shs=(5, 16, 3)), "cpu")
    # 15 + 15 + 20 + 5 + 240 floats, every segment padded to a 16-byte boundary: 16 + 16 + 20 + 8 + 240
    assert ex.numel == 300 and ex.nbytes == 1200
    assert all(v.data_ptr() % 16 == 0 for v in ex.views.values())
    ex.views["shs"].fill_(2.0)
    assert float(ex.flat.sum()) == 2.0 * 5 * 48
    assert ex.views["means3D"].data_ptr() == ex.flat.data_ptr()
    ex.all_reduce()  # no process group: identity


@pytest.mark.timeout(120)
@pytest.mark.parametrize("reduce,P", [("allreduce", 257), ("direct", 257), ("direct", 64)])
def test_allreduce_sums_per_view_gradients_gloo(reduce, P):
    """Both reduction plans: the backend's all-reduce, and the direct form (all-to-all of shards + local sum +
    all-gather), with a length that does and does not divide by the world size."""
    world, K = 2, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, K, q, reduce)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
    expect = {k: torch.zeros(shapes[k]) for k in PARAM_ORDER}
    for r in range(world):  # single-process accumulation of the same per-view gradients
        g = torch.Generator().manual_seed(100 + r)
        for k in PARAM_ORDER:
            expect[k] += torch.randn(shapes[k], generator=g)
    for r in range(world):
        for k in PARAM_ORDER:
            torch.testing.assert_close(torch.from_numpy(res[r][k]), expect[k], rtol=1e-6, atol=1e-6)
    for k in PARAM_ORDER:
        assert (res[0][k] == res[1][k]).all()          # every rank ends with the same bits


# ---- factored SH exchange: all-gather of dRGB + all-reduce of the 11 dense floats ----------------

def torch_sh_reducer(ex):
    """Test-side stand-in for csrc/view_exchange.hip (the product reducer is HIP-only)."""
    from frosting_amd.sh import sh_basis
    P, K = ex.shapes["shs"][:2]
    out = torch.zeros(P, K, 3)
    for v in range(ex.gathered.shape[0]):
        row = ex.gathered[v]
        drgb, campos = row[: 3 * P].view(P, 3), row[3 * P: 3 * P + 3]
        d = ex.means3D - campos
        basis = sh_basis(ex.sh_degree, d / d.norm(dim=1, keepdim=True))
        out[:, : basis.shape[1]] += basis[:, :, None] * drgb[:, None, :]
    ex.views["shs"].copy_(out)


def torch_row_packer(ex):
    """Test-side stand-ins for csrc/view_exchange.hip's pack / scatter kernels (the product ones are HIP-only)."""
    v = ex.views
    dense = torch.cat([v["means3D"], v["scales"], v["opacities"], v["rotations"], ex.own_drgb], 1)     # the row layout, [P, 14]
    idx = ((dense != 0) | dense.isnan()).any(1).nonzero().flatten()
    ex.count_dev.fill_(idx.numel())
    m = min(idx.numel(), ex.rows_own.shape[0])
    ex.rows_own[:m, 0] = idx[:m].to(torch.int32).view(torch.float32)
    ex.rows_own[:m, 1:15] = dense[idx[:m]]
    ex.rows_own[:m, 15] = 0.0


def torch_row_scatterer(ex, rows, n, drgb_dense):
    r = rows[:n]
    idx = r[:, 0].contiguous().view(torch.int32).long()
    v = ex.views
    v["means3D"].index_add_(0, idx, r[:, 1:4])
    v["scales"].index_add_(0, idx, r[:, 4:7])
    v["opacities"].index_add_(0, idx, r[:, 7:8])
    v["rotations"].index_add_(0, idx, r[:, 8:12])
    drgb_dense.view(-1, 3)[idx] = r[:, 12:15]


def _view_inputs(rank, P, K, deg):
    """Per-view gradient of rank `rank`: random dense part, rank-one SH part."""
    from frosting_amd.sh import sh_basis
    g = torch.Generator().manual_seed(7)
    means = torch.randn(P, 3, generator=g)
    g = torch.Generator().manual_seed(200 + rank)
    campos = 4.0 * torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0)
    drgb = torch.randn(P, 3, generator=g)
    drgb[torch.rand(P, generator=g) < 0.3] = 0.0            # culled / clamped rows
    dense = {k: torch.randn(s, generator=g) for k, s in
             dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1)).items()}
    d = means - campos
    basis = sh_basis(deg, d / d.norm(dim=1, keepdim=True))
    shs = torch.zeros(P, K, 3)
    shs[:, : basis.shape[1]] = basis[:, :, None] * drgb[:, None, :]
    return means, campos, drgb, dense, shs


def _factored_worker(rank, world, port, P, K, deg, q, reduce="allreduce"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
    ex = GradientExchange(shapes, "cpu", dist.group.WORLD, factor_sh=True, sh_reducer=torch_sh_reducer, reduce=reduce)
    means, campos, drgb, dense, shs = _view_inputs(rank, P, K, deg)
    ex.set_sh_context(means, deg)
    for k, v in dense.items():
        ex.views[k].copy_(v)
    ex.views["shs"].copy_(shs)                   # what the backward leaves there; replaced by the rebuild
    ex.own_drgb.copy_(drgb)
    ex.own_campos.copy_(campos)
    # 11 dense floats per Gaussian (segments padded to 16 bytes) + the all-gather payload
    assert 11 * P + 3 * P + 4 <= ex.wire_floats_per_rank == ex.layout["shs"][0] + 3 * P + 4 <= 11 * P + 12 + 3 * P + 4
    ex.start()
    ex.wait()
    q.put((rank, {k: ex.views[k].numpy().copy() for k in PARAM_ORDER}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("deg,K,reduce", [(3, 16, "allreduce"), (1, 16, "allreduce"), (3, 16, "direct")])
def test_factored_sh_exchange_equals_sum_of_view_gradients_gloo(deg, K, reduce):
    world, P = 2, 193
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_factored_worker, args=(r, world, port, P, K, deg, q, reduce)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    expect = None
    for r in range(world):
        _, _, _, dense, shs = _view_inputs(r, P, K, deg)
        cur = dict(dense, shs=shs)
        expect = cur if expect is None else {k: expect[k] + cur[k] for k in cur}
    for r in range(world):
        for k in PARAM_ORDER:
            torch.testing.assert_close(torch.from_numpy(res[r][k]), expect[k], rtol=1e-6, atol=1e-6)
    assert (res[0]["shs"] == res[1]["shs"]).all()          # rebuilt identically on every rank


def test_factored_exchange_needs_gpu_reducer():
    ex = GradientExchange(dict(means3D=(4, 3), scales=(4, 3), rotations=(4, 4), opacities=(4, 1), shs=(4, 16, 3)),
                          "cpu", factor_sh=True)
    ex.set_sh_context(torch.zeros(4, 3), 3)
    ex.gathered = torch.zeros(1, ex.payload_numel)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ex.sh_reducer(ex)


# ---- a training step with a parameter update between views --------------------------------------------------
# (exchange -> Adam -> next forward) on N ranks must equal the single-process accumulation of the same N views, bit
# for bit: the in-step schedule leaves nothing of step k in flight when step k+1 reads the parameters.

def _view_gradient(params, view, deg, K, unreached=False):
    """A deterministic stand-in for one view's backward: depends on the CURRENT parameters (so a stale exchange would
    show), dense part arbitrary, SH part rank one per Gaussian as the rasterizer's is (basis(view dir) x dRGB).
    unreached: two Gaussians in three get no gradient at all in this view (no pixel reached them), as on a saturating frame."""
    from frosting_amd.sh import sh_basis
    P = params["means3D"].shape[0]
    campos = 4.0 * torch.nn.functional.normalize(torch.tensor([1.0 + view, 0.5 - view, 2.0]), dim=0)
    dense = {k: torch.sin(params[k] * (1.5 + view)) * (0.1 + 0.01 * view) for k in ("means3D", "scales", "rotations", "opacities")}
    drgb = torch.cos(params["shs"][:, 0, :] * (2.0 + view)) * 0.05
    drgb[(torch.arange(P) + view) % 4 == 0] = 0.0                      # culled / clamped rows of this view
    if unreached:
        dead = (torch.arange(P) * 7 + 3 * view) % 3 != 0
        for t in dense.values():
            t[dead] = 0.0
        drgb[dead] = 0.0
    d = params["means3D"] - campos
    basis = sh_basis(deg, d / d.norm(dim=1, keepdim=True))
    shs = torch.zeros(P, K, 3)
    shs[:, : basis.shape[1]] = basis[:, :, None] * drgb[:, None, :]
    return campos, drgb, dense, shs


def _initial_params(P, K):
    g = torch.Generator().manual_seed(321)
    shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
    return shapes, {k: torch.randn(s, generator=g) for k, s in shapes.items()}


def _train_worker(rank, world, port, P, K, deg, steps, factored, reduce, q, sparse=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes, init = _initial_params(P, K)
    params = {k: v.clone().requires_grad_(True) for k, v in init.items()}
    opt = torch.optim.Adam([params[k] for k in PARAM_ORDER], lr=0.01, eps=1e-15)
    ex = GradientExchange(shapes, "cpu", dist.group.WORLD, factor_sh=factored, sh_reducer=torch_sh_reducer, reduce=reduce,
                          sparse=sparse, row_packer=torch_row_packer, row_scatterer=torch_row_scatterer)
    for it in range(steps):
        with torch.no_grad():
            campos, drgb, dense, shs = _view_gradient({k: v.detach() for k, v in params.items()}, rank, deg, K, unreached=sparse)
            for k, v in dense.items():
                ex.views[k].copy_(v)
            ex.views["shs"].copy_(shs)
            if factored:
                ex.set_sh_context(params["means3D"].detach(), deg)
                ex.own_drgb.copy_(drgb)
                ex.own_campos.copy_(campos)
        ex.start()
        ex.finish_in_step()                  # complete here: nothing is waited for in a later step
        assert not ex._works
        if sparse:                           # a third of the rows travelled (64 B each) instead of all of them (56 B each)
            assert 0 < max(ex.counts) <= P // 3 + 1
            assert ex.wire_floats_per_rank < (11 + 3) * P // 2
        for k in PARAM_ORDER:
            params[k].grad = ex.views[k].clone()
        opt.step()
    q.put((rank, {k: params[k].detach().numpy().copy() for k in PARAM_ORDER}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("factored,reduce", [(False, "allreduce"), (True, "allreduce"), (True, "direct"), (True, "sparse")])
def test_exchange_then_adam_then_next_forward_equals_single_process_accumulation_gloo(factored, reduce):
    """... and the sparse plan (rows of the Gaussians with a gradient: counts, padded all-gather, scatter-add in view
    order, SH rebuild), on views in which two Gaussians in three have no gradient."""
    world, P, K, deg, steps = 2, 131, 16, 3, 3
    sparse = reduce == "sparse"
    reduce = "allreduce" if sparse else reduce
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, P, K, deg, steps, factored, reduce, q, sparse)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # one process: the same views accumulated in view order, then the same optimizer
    shapes, init = _initial_params(P, K)
    params = {k: v.clone().requires_grad_(True) for k, v in init.items()}
    opt = torch.optim.Adam([params[k] for k in PARAM_ORDER], lr=0.01, eps=1e-15)
    for it in range(steps):
        acc = None
        with torch.no_grad():
            for v in range(world):
                _, _, dense, shs = _view_gradient({k: t.detach() for k, t in params.items()}, v, deg, K, unreached=sparse)
                cur = dict(dense, shs=shs)
                acc = cur if acc is None else {k: acc[k] + cur[k] for k in cur}
        for k in PARAM_ORDER:
            params[k].grad = acc[k].clone()
        opt.step()
    for r in range(world):
        for k in PARAM_ORDER:
            assert (torch.from_numpy(res[r][k]) == params[k].detach()).all(), (r, k)     # bit for bit


# ---- slot-sum exchange (round 6): the nine per-Gaussian sums of phase 1 travel, every rank runs the chain for every view ----------
# Stand-ins for csrc/slot_exchange.hip (the product kernels are HIP-only) that write and read the SAME packet layout: header, one
# bit per Gaussian, one row offset per 64 Gaussians, 48-byte rows in index order.

def _sum_chain(params, rows, campos, S):
    """A deterministic stand-in for the per-Gaussian backward chain: the 59 gradient floats of the Gaussians `rows` in one view
    from their twelve sums S [k, 12] (the packet's row), the parameters and the view's camera centre.  Multiplications and
    additions only (the same bits whatever the batch shape); zero sums give a zero row."""
    m, sc, q, o, sh = (params[k][rows] for k in ("means3D", "scales", "rotations", "opacities", "shs"))
    d = m - campos
    w = S[:, 3:6] * d + S[:, 6:9]
    return dict(means3D=w * sc + S[:, 0:3] * d + S[:, 9:12],
                scales=S[:, 3:6] * sc * sc + S[:, 6:9] * d,
                rotations=q * (S[:, 3:4] * d[:, 0:1] + S[:, 8:9]) + S[:, 4:5] * o,
                opacities=S[:, 8:9] * o + S[:, 5:6] * d[:, 1:2],
                shs=sh * 0.0 + (S[:, None, 0:3] * (d[:, None, :] + 0.25)) * torch.arange(1, sh.shape[1] + 1, dtype=torch.float32)[None, :, None])


def _view_sums(params, view, it, P):
    """Phase 1 of view `view` at iteration `it`: twelve sums per Gaussian, zero rows for the Gaussians no pixel reached -- one in
    four in the first two iterations, three in four from the third (the packets sized from the earlier steps then overflow)."""
    campos = torch.tensor([1.0 + view, 0.5 - view, 2.0]) * 0.5
    base = torch.cat([params["means3D"], params["scales"], params["rotations"][:, :3], params["shs"][:, 1, :]], 1)     # [P, 12]
    S = base * (0.25 + 0.125 * view) + 0.0625 * (1 + it)
    reached = (torch.arange(P) * 7 + 3 * view) % 4 == 0 if it < 2 else (torch.arange(P) * 7 + 3 * view) % 4 != 0
    S[~reached] = 0.0
    return campos, S


def torch_sum_packer(ex, c, dest):
    import numpy as np
    from frosting_amd.parallel import SUM_HDR_WORDS, SUM_ROW_FLOATS, sum_packet_words
    first, n = ex.chunks[c]
    cap = ex.capacity[c]
    S = ex.view_ctx["sums"][first:first + n]
    live = (S != 0).any(1).numpy()
    nblk = (n + 63) // 64
    bits = np.zeros(nblk * 64, dtype=bool)
    bits[:n] = live
    masks = np.packbits(bits.reshape(nblk, 64), axis=1, bitorder="little").view(np.uint64).reshape(nblk)
    counts = bits.reshape(nblk, 64).sum(1)
    words = np.zeros(sum_packet_words(n, cap), dtype=np.int32)
    assert words.size == dest.numel()
    want = int(live.sum())
    words[0:6] = [min(want, cap), want, n, cap, first, 0x46534d36]
    words[40:43] = ex.view_ctx["campos"].numpy().view(np.int32)
    words[SUM_HDR_WORDS:SUM_HDR_WORDS + 2 * nblk] = masks.view(np.int32)
    b_at = SUM_HDR_WORDS + 2 * nblk
    words[b_at:b_at + nblk] = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    r_at = (b_at + nblk + 3) // 4 * 4
    rows = S.numpy()[live][:cap]
    words[r_at:r_at + SUM_ROW_FLOATS * rows.shape[0]] = rows.reshape(-1).view(np.int32)
    dest.copy_(torch.from_numpy(words))


def torch_sum_combiner(ex, c, packets, n_views, seq):
    import numpy as np
    from frosting_amd.parallel import SUM_HDR_WORDS, SUM_ROW_FLOATS
    first, n = ex.chunks[c]
    nblk = (n + 63) // 64
    b_at = SUM_HDR_WORDS + 2 * nblk
    r_at = (b_at + nblk + 3) // 4 * 4
    acc = {k: torch.zeros((n,) + tuple(ex.shapes[k][1:])) for k in PARAM_ORDER}
    over, wants = False, []
    for v in range(n_views):                                   # view order
        w = packets[v].numpy()
        assert int(w[2]) == n and int(w[4]) == first and int(w[5]) == 0x46534d36
        want, cap = int(w[1]), int(w[3])
        wants.append(want)
        over |= want > cap
        masks = w[SUM_HDR_WORDS:b_at].view(np.uint64)
        bits = np.unpackbits(masks.view(np.uint8).reshape(nblk, 8), axis=1, bitorder="little").reshape(-1)[:n].astype(bool)
        bases = w[b_at:b_at + nblk]
        idx = np.nonzero(bits)[0]
        # every Gaussian finds its row as base[block] + popcount(bits below it): here for all of them at once
        rank_in_block = (np.cumsum(bits.reshape(-1)) - 1)[idx] - np.concatenate([[0], np.cumsum(bits[: nblk * 64 if n == nblk * 64 else n].astype(np.int64))])[idx // 64 * 64]
        row = bases[idx // 64] + rank_in_block
        ok = row < cap
        idx, row = idx[ok], row[ok]
        S = torch.from_numpy(w[r_at:].view(np.float32)[: SUM_ROW_FLOATS * cap].reshape(cap, SUM_ROW_FLOATS)[row].copy())
        campos = torch.from_numpy(w[40:43].view(np.float32).copy())
        rows = torch.from_numpy(idx).long()
        g = _sum_chain(ex.params, rows + first, campos, S)
        for k in PARAM_ORDER:
            acc[k][rows] += g[k]                               # (indices are unique within a view)
    for k in PARAM_ORDER:
        ex.views[k][first:first + n] = acc[k]
    st = ex.status[c]              # 64-bit words, sequence number << 32 | value: [0] overflow, [1 + v] the rows view v wanted
    st[0] = (seq << 32) | int(over)
    st[1:1 + n_views] = torch.tensor([(seq << 32) | w_ for w_ in wants], dtype=torch.int64)


def _slotsum_worker(rank, world, port, P, K, steps, chunks, q):
    import torch.distributed as dist
    from frosting_amd.parallel import SlotSumExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes, init = _initial_params(P, K)
    params = {k: v.clone().requires_grad_(True) for k, v in init.items()}
    opt = torch.optim.Adam([params[k] for k in PARAM_ORDER], lr=0.01, eps=1e-15)
    ex = SlotSumExchange(shapes, "cpu", dist.group.WORLD, chunks=chunks, packer=torch_sum_packer, combiner=torch_sum_combiner)
    ex.set_params({k: v.detach() for k, v in params.items()})
    caps = []
    for it in range(steps):
        with torch.no_grad():
            campos, S = _view_sums({k: v.detach() for k, v in params.items()}, rank, it, P)
        ex.note_view(sums=S, campos=campos)
        ex.start()
        ex.finish_in_step()                  # complete here, the re-pack of an overflowing chunk included
        assert not ex._works
        caps.append((list(ex.capacity), ex.stats["repacks"]))
        for k in PARAM_ORDER:
            params[k].grad = ex.views[k].clone()
        opt.step()
    q.put((rank, {k: params[k].detach().numpy().copy() for k in PARAM_ORDER}, caps))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world,chunks", [(2, 1), (3, 2), (8, 2)])
def test_slot_sum_exchange_then_adam_equals_single_process_accumulation_gloo(world, chunks):
    """SlotSumExchange on N ranks -- pack the view's rows at the capacity the previous step's views called for, all-gather the
    packets chunk by chunk, run the chain for every view's rows in view order -- with Adam between the views equals the
    single-process accumulation of the same views bit for bit; the packets shrink after the first step (every Gaussian has
    room in it), and the step in which three times as many Gaussians have a gradient overflows them: every rank packs,
    gathers and combines those chunks again, and the result is still the same bits."""
    P, K, steps = 2500, 16, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slotsum_worker, args=(r, world, port, P, K, steps, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=200) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    res = {r: prm for r, prm, _ in got}
    caps = {r: c for r, _, c in got}
    shapes, init = _initial_params(P, K)
    params = {k: v.clone().requires_grad_(True) for k, v in init.items()}
    opt = torch.optim.Adam([params[k] for k in PARAM_ORDER], lr=0.01, eps=1e-15)
    allrows = torch.arange(P)
    for it in range(steps):
        acc = {k: torch.zeros(shapes[k]) for k in PARAM_ORDER}
        with torch.no_grad():
            cur = {k: t.detach() for k, t in params.items()}
            for v in range(world):
                campos, S = _view_sums(cur, v, it, P)
                g = _sum_chain(cur, allrows, campos, S)         # the dense per-view gradient (zero rows where the sums are zero)
                for k in PARAM_ORDER:
                    acc[k] += g[k]
        for k in PARAM_ORDER:
            params[k].grad = acc[k].clone()
        opt.step()
    for r in range(world):
        for k in PARAM_ORDER:
            assert (torch.from_numpy(res[r][k]) == params[k].detach()).all(), (r, k)     # bit for bit
    assert all(caps[r] == caps[0] for r in range(world))               # every rank took the same decisions
    sizes = [sum(c) for c, _ in caps[0]]
    repacks = [n for _, n in caps[0]]
    assert sizes[0] < 0.5 * P and repacks[:2] == [0, 0]                # after step 0 the packets hold a quarter of the Gaussians + slack
    assert repacks[2] == len(caps[0][0][0]) and sizes[2] > sizes[1]    # step 2: every chunk overflowed once and was packed again


def _densify_worker(rank, world, port, P, steps, q):
    import torch.distributed as dist
    from frosting_amd.parallel import DensificationStats
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = DensificationStats(P, "cpu", dist.group.WORLD)
    for it in range(steps):
        g = torch.Generator().manual_seed(1000 * it + rank)
        radii = torch.randint(-2, 30, (P,), generator=g, dtype=torch.int32).clamp_min(0)
        grad = torch.randn(P, 3, generator=g)
        st.update(radii, grad)
    q.put((rank, (st.max_radii2D.numpy().copy(), st.xyz_gradient_accum.numpy().copy(), st.denom.numpy().copy())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_densification_statistics_over_a_view_parallel_batch_gloo():
    """radii MAX and viewspace-gradient-norm / visibility-count SUM across ranks == the reference's per-view updates
    (gaussian_model.py:404-407, train.py:116-117) applied view after view in one process."""
    world, P, steps = 2, 97, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_densify_worker, args=(r, world, port, P, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    max_r, accum, denom = torch.zeros(P), torch.zeros(P, 1), torch.zeros(P, 1)
    for it in range(steps):
        for v in range(world):
            g = torch.Generator().manual_seed(1000 * it + v)
            radii = torch.randint(-2, 30, (P,), generator=g, dtype=torch.int32).clamp_min(0)
            grad = torch.randn(P, 3, generator=g)
            vis = radii > 0
            max_r[vis] = torch.max(max_r[vis], radii[vis].float())                      # train.py:116
            accum[vis] += torch.norm(grad[vis, :2], dim=-1, keepdim=True)              # gaussian_model.py:405-406
            denom[vis] += 1
    for r in range(world):
        assert (torch.from_numpy(res[r][0]) == max_r).all()
        torch.testing.assert_close(torch.from_numpy(res[r][1]), accum, rtol=1e-6, atol=1e-6)
        assert (torch.from_numpy(res[r][2]) == denom).all()


def test_predicted_exchange_budget_arithmetic():
    """frosting_amd.parallel.predict_exchange (the 8-GPU budget bench.py prints as `predicted`): the bytes
    of each plan, the wire times they imply at the nominal xGMI link rate, and the orderings the design relies on."""
    from frosting_amd.parallel import predict_exchange
    P, K = 3_000_000, 16
    plain = predict_exchange(P, K, 8, 1.44, "allreduce", "allreduce", "sync", link_efficiency=1.0)
    assert abs(plain["dense_MB"] - 708.0) < 1e-6 and abs(plain["dense_wire_ms"] - 2 * 7 / 8 * 708e6 / 153e9 * 1e3) < 1e-9   # 8.1 ms
    fac = predict_exchange(P, K, 8, 1.44, "factored", "direct", "sync", link_efficiency=1.0)
    assert abs(fac["dense_MB"] - 132.0) < 1e-6 and abs(fac["gather_MB_in"] - 252.0) < 1e-6
    assert abs(fac["dense_wire_ms"] - 2 * 7 / 8 * 132e6 / (7 * 153e9) * 1e3) < 1e-9
    assert abs(fac["gather_wire_ms"] - 36e6 / 153e9 * 1e3) < 1e-9
    ovl = predict_exchange(P, K, 8, 1.44, "factored", "direct", "in-step", link_efficiency=1.0)
    assert ovl["exposed_ms"] < fac["exposed_ms"] < plain["exposed_ms"]
    assert ovl["scaling_vs_1gpu"] > 6.0 > plain["scaling_vs_1gpu"]
    one = predict_exchange(P, K, 1, 1.44)
    assert one["exposed_ms"] == 0.0 and one["scaling_vs_1gpu"] == 1.0
    # every collective priced at one bus bandwidth per GPU (7 x 153 nominal, 450 = what RCCL usually reaches, 300):
    # incoming bytes / bandwidth, the all-reduce twice
    for bw in (1071.0, 450.0, 300.0):
        f = predict_exchange(P, K, 8, 1.44, "factored", "direct", "in-step", bus_GBps=bw)
        assert abs(f["dense_wire_ms"] - 2 * 7 / 8 * 132e6 / (bw * 1e9) * 1e3) < 1e-9 and abs(f["gather_wire_ms"] - 7 * 36e6 / (bw * 1e9) * 1e3) < 1e-9
        sp = predict_exchange(P, K, 8, 1.44, "sparse", "direct", "sync", bus_GBps=bw)
        assert abs(sp["rows_MB_in"] - 7 * 0.124 * P * 64 / 1e6) < 1e-6 and abs(sp["rows_wire_ms"] - sp["rows_MB_in"] * 1e6 / (bw * 1e9) * 1e3) < 1e-9
        assert sp["dense_MB"] == 0.0 and sp["exposed_ms"] > sp["rows_wire_ms"]
        sa = f["sharded_adam"]      # the sharded optimizer: (N-1)/N of ALL parameters gathered against (N-1)/N of the update saved
        assert abs(sa["param_gather_ms"] - 7 / 8 * 708e6 / (bw * 1e9) * 1e3) < 1e-9 and abs(sa["net_ms"] - (sa["param_gather_ms"] - 0.7)) < 1e-9
    # what the arithmetic says: at the nominal bus the factored in-step plan scales best (> 6 x); the sparse rows
    # overtake it as the bandwidth drops -- and nothing reaches 6 x at 450 GB/s
    at = lambda pl, sc, bw: predict_exchange(P, K, 8, 1.44, pl, "direct", sc, bus_GBps=bw)["scaling_vs_1gpu"]
    assert at("factored", "in-step", 1071.0) > 6.0 > at("sparse", "sync", 1071.0)
    assert at("sparse", "sync", 300.0) > at("factored", "in-step", 300.0)
    assert max(at("factored", "in-step", 450.0), at("sparse", "sync", 450.0)) < 6.0
    for n in (2, 4, 8):           # more GPUs never cost less wire time per rank, and scaling stays below N
        r = predict_exchange(P, K, n, 1.44, "factored", "direct", "in-step")
        assert 0 < r["scaling_vs_1gpu"] < n


def test_predicted_slot_sum_plan_from_measured_local_terms():
    """The slot-sum plan in predict_exchange: the packets' bytes, the two-stage pipeline of gather and combine
    pass over the chunks -- a pass priced x overlap_slowdown while a later chunk is still on the wire --, local terms = the
    single-GPU measurements (SLOTSUM_LOCAL_MS: profiles/r06_combine_bench.log, r06_combine_beside_a_copy.log) -- and what the
    arithmetic then says at 450 GB/s of bus bandwidth."""
    from frosting_amd.parallel import SLOTSUM_LOCAL_MS, predict_exchange, sum_packet_words
    P, render = 3_000_000, 1.35
    loc = SLOTSUM_LOCAL_MS
    for bw in (1071.0, 450.0, 300.0):
        for K in (1, 2, 4):
            r = predict_exchange(P, 16, 8, render, "slotsum", "allgather", "in-step", bus_GBps=bw, rows_fraction=0.1267, chunks=K)
            s = r["slotsum"]
            cap = (int(0.1267 * P * loc["slack"]) // 256 + 1) * 256
            assert s["capacity_rows"] == cap and abs(s["packet_MB"] * 1e6 - (4.0 * sum_packet_words(P, cap) + 256.0 * (K - 1))) < 1.0
            assert abs(s["wire_ms"] - 7 * s["packet_MB"] * 1e6 / (bw * 1e9) * 1e3) < 1e-9
            c_all = loc["combine_ms"][8] + loc["per_chunk_ms"] * (K - 1)
            assert abs(s["combine_ms"] - c_all) < 1e-12
            # the pipeline, restated: gather k ends at (k + 1) W / K; pass k starts when its packets are there and pass k - 1 is done
            w1, c1, t = s["wire_ms"] / K, c_all / K, 0.0
            for k in range(K):
                now, left = max(t, (k + 1) * w1), c1
                if now < s["wire_ms"]:
                    beside = min(left * loc["overlap_slowdown"], s["wire_ms"] - now)
                    left -= beside / loc["overlap_slowdown"]
                    now += beside
                t = now + left
            assert abs(s["gather_and_combine_ms"] - t) < 1e-12
            if K == 1:
                assert abs(t - (s["wire_ms"] + c_all)) < 1e-12          # one chunk: nothing overlaps, nothing is slowed
            assert s["wire_ms"] + c1 - 1e-12 <= t <= s["wire_ms"] + c_all * loc["overlap_slowdown"] + 1e-12
            assert abs(r["exposed_ms"] - (s["pack_ms"] + t - loc["phase2_ms"])) < 1e-12
            assert r["dense_MB"] == 0.0
    at = lambda bw, K: predict_exchange(P, 16, 8, render, "slotsum", "allgather", "in-step", bus_GBps=bw, rows_fraction=0.1267, chunks=K)["scaling_vs_1gpu"]
    assert at(450.0, 4) >= 6.0 and at(450.0, 2) >= 6.0 > at(450.0, 1)        # the chunks' overlap is what carries it past 6 x
    assert at(450.0, 2) > predict_exchange(P, 16, 8, render, "factored", "direct", "in-step", bus_GBps=450.0)["scaling_vs_1gpu"]
    assert at(1071.0, 2) >= 6.0 > at(300.0, 4)                               # 300 GB/s: wire-bound, below the target
    for n in (2, 4, 8):
        assert 0 < predict_exchange(P, 16, n, render, "slotsum", "allgather", "in-step", bus_GBps=450.0, rows_fraction=0.1267)["scaling_vs_1gpu"] < n


@pytest.mark.timeout(300)
def test_bench_self_launch_at_world_8_dry_run():
    """`python bench.py --gpus 8` from a bare shell: the self-launch under torch.distributed.run, eight ranks'
    environment, the group, the slot-sum exchange's collectives on toy packets in the real layout, the barrier + max-over-ranks
    timing and rank 0's one JSON line -- without a GPU (--dry-run-plumbing; the line is marked dry_run, its value is no measurement)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--dry-run-plumbing",
                        "--steps", "3", "--warmup", "1", "--points", "20000"], cwd="/tmp", env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]             # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["ranks"] == 8 and d["dry_run"] is True and d["all_ranks_agree"] is True
    assert d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["config"]["exchange"] == "slotsum"
    assert d["config"]["chunks"] == 2 and d["config"]["repacks"] == 0
    assert abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


def _probe_worker(rank, world, port, q):
    import torch.distributed as dist
    from frosting_amd.parallel import probe_reduce_plan
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = dict(means3D=(301, 3), scales=(301, 3), rotations=(301, 4), opacities=(301, 1), shs=(301, 16, 3))
    exs = [GradientExchange(shapes, "cpu", dist.group.WORLD, factor_sh=f, sh_reducer=lambda ex: None) for f in (True, True)]
    times, best = probe_reduce_plan(exs, iters=2, warm=1)
    # ... and the exchange still works afterwards, with the chosen plan
    for ex in exs:
        ex.flat.fill_(float(rank + 1))
    exs[0].means3D = torch.zeros(301, 3)
    exs[0].start()
    exs[0].wait()
    q.put((rank, times, best, exs[0].reduce, exs[1].reduce, float(exs[0].views["means3D"][0, 0])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_probe_of_the_reduce_plans_gloo():
    """bench.py --reduce auto: frosting_amd.parallel.probe_reduce_plan times the dense sum both ways on live exchange
    buffers and sets every exchange to the faster plan -- the same plan on every rank (the times are reduced first), and
    the exchange sums correctly afterwards."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_probe_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (_, t0, b0, r00, r01, v0), (_, t1, b1, r10, r11, v1) = sorted(res)
    assert t0 == t1 and b0 == b1 and b0 in ("allreduce", "direct") and set(t0) == {"allreduce", "direct"}
    assert r00 == r01 == r10 == r11 == b0
    assert v0 == v1 == 3.0                       # 1 + 2: the dense part summed over the two ranks


# ---- sharded optimizer (second option): reduce-scatter -> update of the 1/N shard -> all-gather of the parameters

def _torch_shard_step(opt, lo, params, grads, lrs, head_lrs, grad_scale):
    """Test-side stand-in for frg_adam_step_shard: plain Adam on the elements [lo, lo + len) of the flat layout, the step size
    of every element taken from its GLOBAL position (segment, and DC / rest phase inside the SH rows)."""
    n = params.numel()
    idx = torch.arange(lo, lo + n)
    lr = torch.zeros(n)
    begin = 0
    for k, name in enumerate(opt.names):
        end = int(opt._ends[k])
        sel = (idx >= begin) & (idx < end)
        lr[sel] = lrs[k]
        if opt._period[k] > 0:
            head = sel & (((idx - begin) % int(opt._period[k])) < int(opt._head[k]))
            lr[head] = head_lrs[k]
        begin = end
    b1, b2 = opt.betas
    t = opt.steps + 1
    g = grads * grad_scale
    opt.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
    opt.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = opt.exp_avg_sq.sqrt() / (1 - b2 ** t) ** 0.5 + opt.eps
    params.sub_(lr / (1 - b1 ** t) * (opt.exp_avg / denom))


def _sharded_worker(rank, world, port, P, K, steps, q):
    import torch.distributed as dist
    from frosting_amd.optim import ShardedFlatAdam
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes, init = _initial_params(P, K)
    lrs = dict(means3D=1.6e-4, scales=5e-3, rotations=1e-3, opacities=5e-2, shs=2.5e-3 / 20)
    opt = ShardedFlatAdam(shapes, lrs, "cpu", dist.group.WORLD, sh_dc_lr=2.5e-3, shard_step=_torch_shard_step)
    for k in PARAM_ORDER:
        opt.params[k].copy_(init[k])
    assert opt.exp_avg.numel() == opt.shard and opt.shard * world >= opt.numel      # the moments exist for the shard only
    for it in range(steps):
        g = torch.Generator().manual_seed(1000 * it + rank)
        opt.step(torch.randn(opt.numel, generator=g))
        if it == 0:      # the reference loop's opacity reset (replace_tensor_to_optimizer) + a reset of some rows of another group
            opt.reset("opacities", torch.full((P, 1), 0.01))
            opt.reset("shs", rows=torch.arange(P) % 3 == 0)
    q.put((rank, opt.flat.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_adam_equals_the_replicated_update_gloo(world):
    """ShardedFlatAdam on N ranks (reduce-scatter of the per-view gradients, each rank's update of its shard, all-gather of
    the parameters; the shard boundaries fall inside the SH rows, off the DC / rest period) == the same update of the
    whole buffer on the summed gradients in one process, bit for bit, on every rank -- with the reference loop's opacity
    reset and a reset of selected rows between two steps (ShardedFlatAdam.reset: the values on every rank, the moments where
    the group meets the rank's shard)."""
    P, K, steps = 131, 16, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, P, K, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # one process: the whole buffer as one "shard"
    import types
    from frosting_amd.parallel import flat_layout
    shapes, init = _initial_params(P, K)
    names, layout, numel = flat_layout(shapes)
    ref = types.SimpleNamespace(names=names, betas=(0.9, 0.999), eps=1e-15, steps=0, exp_avg=torch.zeros(numel), exp_avg_sq=torch.zeros(numel))
    ref._ends = [layout[names[i + 1]][0] if i + 1 < len(names) else numel for i in range(len(names))]
    ref._period = [0] * len(names); ref._head = [0] * len(names)
    ref._period[names.index("shs")], ref._head[names.index("shs")] = K * 3, 3
    flat = torch.zeros(numel)
    for k in PARAM_ORDER:
        o, n = layout[k]
        flat[o:o + n] = init[k].reshape(-1)
    lrs = dict(means3D=1.6e-4, scales=5e-3, rotations=1e-3, opacities=5e-2, shs=2.5e-3 / 20)
    for it in range(steps):
        total = None
        for r in range(world):       # rank order, like the all-to-all + sum
            g = torch.randn(numel, generator=torch.Generator().manual_seed(1000 * it + r))
            total = g if total is None else total + g
        _torch_shard_step(ref, 0, flat, total, [lrs[k] for k in names], [2.5e-3 if k == "shs" else 0.0 for k in names], 1.0)
        ref.steps += 1
        if it == 0:      # FlatAdam.reset on the whole buffer: values + both moments of the group / of the selected rows
            o, n = layout["opacities"]
            flat[o:o + n] = 0.01
            ref.exp_avg[o:o + n] = 0.0
            ref.exp_avg_sq[o:o + n] = 0.0
            o, n = layout["shs"]
            rows = torch.arange(P) % 3 == 0
            ref.exp_avg[o:o + n].view(P, K, 3)[rows] = 0.0
            ref.exp_avg_sq[o:o + n].view(P, K, 3)[rows] = 0.0
    for r in range(world):
        assert (torch.from_numpy(res[r]) == flat).all(), r

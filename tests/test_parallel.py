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


def _worker(rank, world, port, P, K, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
    ex = GradientExchange(shapes, "cpu", dist.group.WORLD)
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
    assert ex.numel == 5 * 59 and ex.nbytes == 5 * 236
    ex.views["shs"].fill_(2.0)
    assert float(ex.flat.sum()) == 2.0 * 5 * 48
    assert ex.views["means3D"].data_ptr() == ex.flat.data_ptr()
    ex.all_reduce()  # no process group: identity


@pytest.mark.timeout(120)
def test_allreduce_sums_per_view_gradients_gloo():
    world, P, K = 2, 257, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, K, q)) for r in range(world)]
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

"""View-parallel data parallelism (BASELINE config 5).

The reference is single-GPU (no torch.distributed anywhere).  The path shards
over camera views: parameters are replicated (236 B/Gaussian), rank k renders
view k forward+backward, and the one exchange step is the SUM of the dense
per-Gaussian parameter gradients {means3D[P,3], scales[P,3], rotations[P,4],
opacities[P,1], shs[P,K,3]} -- one flat fp32 buffer, one RCCL all-reduce
(backend "nccl" on ROCm), no other collective on the data path.  A single image
is never split across GPUs.

Correctness oracle: the sum of the single-GPU per-view gradients (tests/test_parallel.py,
world_size-2 gloo on CPU for the exchange; gpu tests for the rasterizer side).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

PARAM_ORDER = ("means3D", "scales", "rotations", "opacities", "shs")


class GradientExchange:
    """Flat fp32 gradient buffer with named per-parameter views and a single
    all-reduce.  Backend-agnostic (RCCL on GPUs, gloo in the CPU tests)."""

    def __init__(self, shapes: dict, device, process_group=None, average: bool = False):
        self.shapes = {k: tuple(shapes[k]) for k in PARAM_ORDER if k in shapes}
        self.device = torch.device(device)
        self.group = process_group
        self.average = average
        sizes = {k: int(torch.Size(s).numel()) for k, s in self.shapes.items()}
        self.numel = sum(sizes.values())
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.views, o = {}, 0
        for k, n in sizes.items():
            self.views[k] = self.flat[o:o + n].view(self.shapes[k])
            o += n

    @property
    def nbytes(self) -> int:
        return self.numel * 4

    def all_reduce(self):
        """Blocking (stream-ordered) SUM over the group; returns the flat buffer."""
        self.start()
        return self.wait()

    def start(self):
        """Enqueue the all-reduce behind the work already on the current stream and return at
        once (torch.distributed async_op): the collective runs on the backend's own stream, so
        kernels enqueued afterwards on the compute stream overlap with it."""
        import torch.distributed as dist
        self._work = None
        if self.group is None or not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return None
        self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return self._work

    def wait(self):
        """Make the current stream (CPU backends: the caller) wait for the pending all-reduce."""
        import torch.distributed as dist
        work = getattr(self, "_work", None)
        if work is not None:
            work.wait()
            self._work = None
            if self.average:
                self.flat.mul_(1.0 / dist.get_world_size(self.group))
        return self.flat


class _Arena:
    """Grow-only device buffer handed to the C ABI's allocation callbacks: after
    the first view no allocator call remains on the per-step path (image size and
    P are constant per model, only R varies -- SURVEY Appendix A-18)."""

    def __init__(self, device, slack: float = 1.25):
        self.device, self.slack = device, slack
        self.buf = torch.empty(0, dtype=torch.uint8, device=device)
        self.cb = _lib.ALLOC_FN(self._alloc)

    def _alloc(self, _user, nbytes):
        if self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes * self.slack) + 256, dtype=torch.uint8, device=self.device)
        return self.buf.data_ptr()

    def ensure(self, nbytes):
        self._alloc(None, nbytes)
        return self.buf


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class ViewParallelRasterizer:
    """Replicated scene + per-rank view render through the C ABI, gradients written
    in place into the flat exchange buffer (no copies, no zero-fill)."""

    def __init__(self, scene, device, process_group=None, average: bool = False):
        self.dev = torch.device(device)
        self.scene = scene
        P, K = scene.means3D.shape[0], scene.shs.shape[1]
        self.P, self.K = P, K
        shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
        # two gradient buffers: the exchange of step k may still be in flight on the
        # collective stream while step k+1 renders and writes the other buffer
        self.exchanges = [GradientExchange(shapes, self.dev, process_group, average) for _ in range(2)]
        self.exchange = self.exchanges[0]
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=self.dev)
        # rank-local (not exchanged) backward outputs
        self.dL_dmeans2D, self.dL_dconic, self.dL_dcolors, self.dL_dcov3D = f(P, 3), f(P, 4), f(P, 3), f(P, 6)
        self.geom, self.binning, self.img, self.work = (_Arena(self.dev) for _ in range(4))
        self.radii = torch.empty(P, dtype=torch.int32, device=self.dev)
        self.out_color = None
        self.num_rendered = 0
        self._view = None

    def forward(self, cam, bg):
        L = _lib.lib()
        s = self.scene
        H, W = cam.image_height, cam.image_width
        if self.out_color is None or tuple(self.out_color.shape) != (3, H, W):
            self.out_color = torch.empty((3, H, W), dtype=torch.float32, device=self.dev)
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        rc = L.frg_forward(self.geom.cb, self.binning.cb, self.img.cb, None,
                           self.P, s.sh_degree, self.K, _p(bg), W, H,
                           _p(s.means3D), _p(s.shs), None, _p(s.opacities),
                           _p(s.scales), 1.0, _p(s.rotations), None,
                           _p(cam.viewmatrix), _p(cam.projmatrix), _p(cam.campos),
                           float(cam.tanfovx), float(cam.tanfovy), 0,
                           _p(self.out_color), _p(self.radii), 0, stream)
        if rc < 0:
            raise RuntimeError(f"frg_forward failed ({rc}): {_lib.last_error()}")
        self.num_rendered = rc
        self._view = (cam, bg)
        return self.out_color, self.radii

    def backward(self, dL_dimage, slot: int = 0):
        """Gradients of the last forward, written in place into exchange buffer `slot`."""
        L = _lib.lib()
        s = self.scene
        cam, bg = self._view
        H, W = cam.image_height, cam.image_width
        self.exchange = self.exchanges[slot]
        g = self.exchange.views
        ws = int(L.frg_backward_workspace_bytes(self.P, self.num_rendered))
        work = self.work.ensure(ws)
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        rc = L.frg_backward(self.P, s.sh_degree, self.K, self.num_rendered, _p(bg), W, H,
                            _p(s.means3D), _p(s.shs), None,
                            _p(s.scales), 1.0, _p(s.rotations), None,
                            _p(cam.viewmatrix), _p(cam.projmatrix), _p(cam.campos),
                            float(cam.tanfovx), float(cam.tanfovy), _p(self.radii),
                            _p(self.geom.buf), _p(self.binning.buf), _p(self.img.buf), _p(dL_dimage),
                            _p(self.dL_dmeans2D), _p(self.dL_dconic), _p(g["opacities"]), _p(self.dL_dcolors),
                            _p(g["means3D"]), _p(self.dL_dcov3D), _p(g["shs"]), _p(g["scales"]), _p(g["rotations"]),
                            _p(work), work.numel(), 0, stream)
        if rc < 0:
            raise RuntimeError(f"frg_backward failed ({rc}): {_lib.last_error()}")
        return g

    def allreduce_grads(self, slot: int = 0):
        """Synchronous form: SUM over ranks, stream-ordered."""
        return self.exchanges[slot].all_reduce()

    def start_exchange(self, slot: int):
        """Launch the all-reduce of buffer `slot` behind its backward; returns immediately."""
        return self.exchanges[slot].start()

    def wait_exchange(self, slot: int):
        """Order the current stream after the pending all-reduce of buffer `slot` (call before
        reading its gradients or before the next backward that overwrites it)."""
        return self.exchanges[slot].wait()

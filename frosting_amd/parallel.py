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
SEGMENT_ALIGN = 4   # floats: every parameter's segment of the flat layout starts on a 16-byte boundary


def flat_layout(shapes: dict, names=None):
    """Offsets of the named tensors inside the flat fp32 buffer shared by the gradient exchange and the
    fused optimizer.  Each segment starts on a 16-byte boundary (the backward stores float4 rows into
    these views); the pad elements belong to the segment before them and stay zero.
    Returns (names, {name: (offset, numel)}, total)."""
    names = list(names) if names is not None else [k for k in PARAM_ORDER if k in shapes]
    out, o = {}, 0
    for k in names:
        n = int(torch.Size(tuple(shapes[k])).numel())
        out[k] = (o, n)
        o = (o + n + SEGMENT_ALIGN - 1) // SEGMENT_ALIGN * SEGMENT_ALIGN
    return names, out, o


def _hip_sh_reducer(ex: "GradientExchange"):
    """dL_dsh <- sum over views of basis(mean - campos_v) (x) dRGB_v (frg_sh_grad_from_views)."""
    if ex.flat.device.type != "cuda":
        raise RuntimeError("the factored SH exchange rebuilds dL_dsh with a HIP kernel: gradients must live on the GPU "
                           "(no CPU path; tests inject their own reducer)")
    L = _lib.lib()
    P, K = ex.shapes["shs"][0], ex.shapes["shs"][1]
    stride = ex.gathered.shape[1]
    stream = C.c_void_p(torch.cuda.current_stream(ex.flat.device).cuda_stream)
    base = ex.gathered.data_ptr()
    rc = L.frg_sh_grad_from_views(P, ex.sh_degree, K, ex.gathered.shape[0], _p(ex.means3D),
                                  C.c_void_p(base + 4 * 3 * P), stride, C.c_void_p(base), stride,
                                  _p(ex.views["shs"]), stream)
    if rc < 0:
        raise RuntimeError(f"frg_sh_grad_from_views failed ({rc}): {_lib.last_error()}")


ROW_FLOATS = 16    # a row of the sparse exchange: index bits, dL_dmeans3D[3], dL_dscales[3], dL_dopacity, dL_drotations[4], dRGB[3], pad


def _hip_row_packer(ex: "GradientExchange"):
    """rows_own <- the non-zero rows of this view's dense gradients + dRGB, count_dev <- their number (frg_pack_grad_rows)."""
    if ex.flat.device.type != "cuda":
        raise RuntimeError("the sparse exchange packs its rows with a HIP kernel: gradients must live on the GPU "
                           "(no CPU path; tests inject their own packer)")
    v = ex.views
    stream = C.c_void_p(torch.cuda.current_stream(ex.flat.device).cuda_stream)
    rc = _lib.lib().frg_pack_grad_rows(ex.P, _p(v["means3D"]), _p(v["scales"]), _p(v["rotations"]), _p(v["opacities"]),
                                       _p(ex.own_drgb), _p(ex.rows_own), ex.rows_own.shape[0], _p(ex.count_dev), stream)
    if rc < 0:
        raise RuntimeError(f"frg_pack_grad_rows failed ({rc}): {_lib.last_error()}")


def _hip_row_scatterer(ex: "GradientExchange", rows: torch.Tensor, n: int, drgb_dense):
    """dense gradient arrays += rows[:n]; drgb_dense[index] <- the rows' dRGB (frg_scatter_grad_rows)."""
    if ex.flat.device.type != "cuda":
        raise RuntimeError("the sparse exchange scatters its rows with a HIP kernel (no CPU path; tests inject their own)")
    v = ex.views
    stream = C.c_void_p(torch.cuda.current_stream(ex.flat.device).cuda_stream)
    rc = _lib.lib().frg_scatter_grad_rows(int(n), ex.P, _p(rows), _p(v["means3D"]), _p(v["scales"]), _p(v["rotations"]),
                                          _p(v["opacities"]), _p(drgb_dense), stream)
    if rc < 0:
        raise RuntimeError(f"frg_scatter_grad_rows failed ({rc}): {_lib.last_error()}")


# ---- the 8-GPU budget -----------------------------------------------------------------------------
XGMI_LINKS = 7                 # one node: every MI355X talks to each of the other seven over its own link
XGMI_LINK_GBPS = 153.0         # per link and direction (the figure the task statement and SURVEY 8(e) give)


# The slot-sum plan's local terms at C3 (3 M Gaussians, 0.127 P rows per view), measured on ONE MI355X with HIP events
# (tools/combine_bench.py: profiles/r06_combine_bench.log): pack = mask scan + rows; combine_ms[N] = frg_backward_combine over
# N views' packets in one chunk; per_chunk_ms = what every further chunk adds (launch + tail); phase2_ms = one-call backward -
# phase 1.  overlap_slowdown: the combine pass beside a stand-in for a collective's kernel (a copy kernel of 8 .. 64 workgroups on a
# side stream: 1.21 .. 1.34 x, profiles/r06_combine_beside_a_copy.log) -- a proxy, RCCL's own kernels were never beside it; slack: a setting.
SLOTSUM_LOCAL_MS = {"pack_ms": 0.036, "pack_per_chunk_ms": 0.005, "combine_ms": {1: 0.168, 2: 0.189, 4: 0.215, 8: 0.293},
                    "per_chunk_ms": 0.011, "phase2_ms": 0.115, "overlap_slowdown": 1.3, "slack": 1.125}


def predict_exchange(P: int, K: int, world: int, render_ms: float, plan: str = "factored", reduce: str = "allreduce",
                     schedule: str = "in-step", link_efficiency: float = 0.8, rebuild_ms_per_view: float = 0.019,
                     split_overhead_ms: float = 0.08, per_gaussian_bwd_ms: float = 0.29, bus_GBps: float = None,
                     rows_fraction: float = 0.124, hbm_GBps: float = 5300.0, adam_ms: float = 0.8, chunks: int = 2,
                     slotsum_local: dict = None):
    """Predicted step time and scaling of the view-parallel step on ONE node of `world` MI355X: arithmetic, not a
    measurement (no 8-GPU run is available to this repository; bench.py prints it as `predicted`).

    Bytes per rank: the dense part is 11 floats per Gaussian (44 P bytes) in the factored plan, (11 + 3 K) floats in the
    plain one; the factored plan adds an all-gather of 3 floats per Gaussian and view; the sparse plan moves ROWS of 64
    bytes for the rows_fraction x P Gaussians with a gradient (0.124 = the 371 447 of 3 M measured at C3).  Wire time of a
    collective over the fully connected xGMI node, when bus_GBps is None -- from the links:
      * direct (reduce-scatter + all-gather of 1/N shards, every link busy): 2 (N-1)/N bytes / (links in use x rate),
        links in use = min(N - 1, 7);
      * RCCL all-reduce is taken as a ring bound by ONE link: 2 (N-1)/N bytes / rate -- the pessimistic reading of
        SURVEY 8(e); RCCL may do better on this topology, which is for the driver's curve to say;
      * all-gather of the payloads / the rows: every peer's block arrives over its own link: block bytes / rate.
    With bus_GBps (the aggregate rate at which ONE GPU receives / sends during a collective -- RCCL's "bus bandwidth";
    7 x 153 = 1071 nominal): every collective moves its incoming bytes, (N-1)/N x the total, at that rate (the all-reduce
    twice: reduce-scatter + all-gather), whatever its algorithm.
    Exposed time (what the step grows by): schedule "in-step" with the two-call backward hides the payload all-gather
    under phase 2 of the backward (per_gaussian_bwd_ms) at the price of split_overhead_ms; the SH rebuild
    (rebuild_ms_per_view x N, measured 0.15 ms at 8 views) runs beside the dense sum; "sync" hides nothing.  Sparse plan
    (one-call backward, nothing hidden but the zero fills, which run under the wire): pack (one read of the 56 P bytes) +
    the host's wait for the counts and its serial section behind it (0.16 ms, measured on one rank) + the row all-gather + one scatter launch per view + the SH rebuild.
    Slot-sum plan (round 6, plan="slotsum"): per rank and step one all-gather per chunk of fixed-capacity packets -- 48-byte rows
    for rows_fraction x P x slack Gaussians, one bit per Gaussian, one word per 64 -- and LOCAL terms that were measured on one
    MI355X (slotsum_local, defaults = SLOTSUM_LOCAL_MS: tools/combine_bench.py, profiles/r06_combine_bench.log): pack_ms, the
    combine pass over `world` views' packets (combine_ms[world], total over the chunks), and phase2_ms = what the per-view
    backward no longer does (the per-Gaussian chain and the dense rows of ONE view: one-call backward - phase 1).  The chunks
    pipeline: chunk k's combine pass runs while chunk k + 1 is on the wire (simulated chunk by chunk); a pass is priced
    overlap_slowdown x its stand-alone time for as long as a later chunk is still travelling (measured with a stand-in copy
    kernel beside it: x 1.3), its stand-alone time afterwards.  exposed = pack + that - phase2.
    link_efficiency: achieved / nominal link rate.  Returns a dict."""
    rate = XGMI_LINK_GBPS * 1e9 * link_efficiency
    links = max(1, min(world - 1, XGMI_LINKS))
    factored = plan in ("factored", "sparse")
    sparse = plan == "sparse"
    bus = bus_GBps * 1e9 if bus_GBps else None
    dense_bytes = 4 * P * (11 if factored else 11 + 3 * K)
    frac = 2.0 * (world - 1) / world if world > 1 else 0.0
    if bus:
        dense_ms = 1e3 * frac * dense_bytes / bus
        gather_ms = 1e3 * (12.0 * P * (world - 1) / bus) if (factored and world > 1) else 0.0
    else:
        dense_ms = 1e3 * frac * dense_bytes / (rate * (links if reduce == "direct" else 1))
        gather_ms = 1e3 * (12.0 * P / rate) if (factored and world > 1) else 0.0
    rebuild_ms = rebuild_ms_per_view * world if factored else 0.0
    rows = rows_fraction * P
    rows_ms = pack_ms = scatter_ms = zero_ms = counts_ms = 0.0
    if sparse and world > 1:
        row_bytes = 4.0 * ROW_FLOATS * rows
        rows_ms = 1e3 * (row_bytes * (world - 1) / bus if bus else row_bytes / rate)
        pack_ms = 1e3 * (56.0 * P + row_bytes) / (hbm_GBps * 1e9)
        # the host's wait for the counts and what it enqueues behind it with the GPU idle: a single rank's exposed time (0.35 ms,
        # profiles/r05_exchange_1rank.log) less the kernels counted here
        counts_ms = 0.16
        zero_ms = 1e3 * (44.0 * P + 12.0 * P * world) / (hbm_GBps * 1e9)      # the dense part and the dense dRGB of every view
        # one scatter launch per view: 0.047 ms for the 372 000 rows of a C3 view (rocprofv3, profiles/r05_trace_c3_sparse_exchange.log)
        scatter_ms = world * 0.047 * rows / 372000.0
    slot = None
    if plan == "slotsum" and world > 1:
        loc = dict(SLOTSUM_LOCAL_MS)
        loc.update(slotsum_local or {})
        K = max(1, int(chunks))
        cap = min(P, (int(rows_fraction * P * loc["slack"]) // 256 + 1) * 256)
        packet_bytes = 4.0 * sum_packet_words(P, cap) + 256.0 * (K - 1)
        wire = 1e3 * (packet_bytes * (world - 1) / bus if bus else packet_bytes / rate)
        table = loc["combine_ms"]                               # {views: ms of one pass over everything}, interpolated
        ks = sorted(table)
        lo_k = max([k for k in ks if k <= world] or ks[:1]); hi_k = min([k for k in ks if k >= world] or ks[-1:])
        c_all = table[lo_k] if hi_k == lo_k else table[lo_k] + (table[hi_k] - table[lo_k]) * (world - lo_k) / (hi_k - lo_k)
        c_all = c_all + loc["per_chunk_ms"] * (K - 1)
        # the two-stage pipeline, chunk by chunk: gather k ends at (k + 1) w; pass k starts when its packets are there and pass
        # k - 1 is done, and takes c -- x overlap_slowdown while a later chunk is still on the wire (a second queue's kernel
        # beside it: measured with a stand-in copy kernel, profiles/r06_combine_beside_a_copy.log), c alone for what remains
        w1, c1 = wire / K, c_all / K
        t, slowed = 0.0, 0.0
        for k in range(K):
            start = max(t, (k + 1) * w1)
            left, now = c1, start
            if now < wire:                                     # some of this pass runs beside the remaining gathers
                beside = min(left * loc["overlap_slowdown"], wire - now)
                left -= beside / loc["overlap_slowdown"]
                now += beside
                slowed += beside
            t = now + left
        done = t
        pack = loc["pack_ms"] + loc["pack_per_chunk_ms"] * (K - 1)
        slot = {"chunks": K, "capacity_rows": cap, "packet_MB": packet_bytes / 1e6, "wire_ms": wire, "combine_ms": c_all, "pack_ms": pack,
                "phase2_saved_ms": loc["phase2_ms"], "gather_and_combine_ms": done, "combine_ms_beside_the_wire": slowed,
                "bound": "wire" if done - wire <= c1 * 1.0001 else "combine", "local_terms": loc}
        dense_ms = gather_ms = rebuild_ms = 0.0
        dense_bytes = 0
    if world == 1:
        exposed = 0.0
    elif slot is not None:
        exposed = slot["pack_ms"] + slot["gather_and_combine_ms"] - slot["phase2_saved_ms"]
    elif sparse:
        exposed = pack_ms + counts_ms + max(rows_ms, zero_ms) + scatter_ms + rebuild_ms
    elif not factored:
        exposed = dense_ms
    elif schedule == "in-step":
        exposed = split_overhead_ms + max(0.0, gather_ms - per_gaussian_bwd_ms) + max(dense_ms, rebuild_ms)
    else:
        exposed = gather_ms + dense_ms + rebuild_ms
    step = render_ms + exposed
    out = {"world": world, "plan": plan, "reduce": reduce, "schedule": schedule, "dense_MB": (0.0 if sparse else dense_bytes / 1e6),
           "gather_MB_in": 12.0 * P * max(world - 1, 0) / 1e6 if (factored and not sparse) else 0.0, "dense_wire_ms": 0.0 if sparse else dense_ms,
           "gather_wire_ms": 0.0 if sparse else gather_ms, "sh_rebuild_ms": rebuild_ms, "exposed_ms": exposed, "ms_per_step": step,
           "scaling_vs_1gpu": world * render_ms / step, "link_GBps": XGMI_LINK_GBPS, "link_efficiency": link_efficiency,
           "bus_GBps": bus_GBps,
           "note": "arithmetic from bytes and the nominal xGMI link rate (or the given bus bandwidth), not a measurement"}
    if slot is not None:
        out["slotsum"] = slot
    if sparse:
        out.update(rows_per_rank=rows, rows_MB_in=4.0 * ROW_FLOATS * rows * max(world - 1, 0) / 1e6, rows_wire_ms=rows_ms,
                   pack_ms=pack_ms, counts_wait_ms=counts_ms, zero_fill_ms=zero_ms, scatter_ms=scatter_ms)
    # SURVEY 8(e), second option (frosting_amd.optim.ShardedFlatAdam): the all-gather half of the sum moves PARAMETERS instead
    # of gradients -- (N-1)/N of all 59 floats per Gaussian, whatever the exchange plan -- and takes (N-1)/N of the replicated
    # Adam (adam_ms: 0.8 ms at C3) off every rank
    if world > 1:
        full = 4.0 * P * (11 + 3 * K) * (world - 1) / world
        gather = 1e3 * full / (bus if bus else rate * links)
        out["sharded_adam"] = {"param_gather_ms": gather, "adam_saved_ms": adam_ms * (world - 1) / world,
                               "net_ms": gather - adam_ms * (world - 1) / world,
                               "note": "net > 0: the sharded optimizer costs more than the replicated one on this node"}
    return out


class GradientExchange:
    """Flat fp32 gradient buffer with named per-parameter views.  Backend-agnostic (RCCL on
    GPUs, gloo in the CPU tests).

    Two exchange plans, both yielding the SUM over ranks of every per-Gaussian gradient:
      * plain:    one all-reduce of the whole buffer (59 floats per Gaussian at SH degree 3);
      * factored: the SH gradient of one view is rank one per Gaussian -- basis(view dir) (x) dRGB,
        backward.cu:20-139 -- and the basis depends on replicated data only, so ranks all-gather
        dRGB (3 floats) and all-reduce the 11 other floats, then rebuild sum_v dL_dsh_v locally
        (csrc/view_exchange.hip).  2.6x fewer bytes over xGMI; identical on every rank and
        independent of the collective's reduction order for the SH part."""

    slotsum = False      # (SlotSumExchange: the plan of round 6)

    def __init__(self, shapes: dict, device, process_group=None, average: bool = False,
                 factor_sh: bool = False, sh_reducer=None, reduce: str = "allreduce", sparse: bool = False,
                 row_packer=None, row_scatterer=None):
        """reduce: how the summed part travels.  "allreduce": one collective, the backend picks the
        algorithm (RCCL: rings / trees over xGMI).  "direct": reduce-scatter written as ONE all-to-all of
        1/N shards -- every GPU sends shard j straight to GPU j over its own xGMI link, all seven links of
        the fully connected node busy at once -- a local sum of the N received shards, then an all-gather of
        the reduced shards (a ring is bound by one link, ~153 GB/s; the direct form by
        seven)."""
        self.shapes = {k: tuple(shapes[k]) for k in PARAM_ORDER if k in shapes}
        self.device = torch.device(device)
        # sparse: the third plan (round 5).  Per view only the Gaussians some pixel reached carry a gradient (one visible
        # Gaussian in seven at C3), so ranks all-gather ROWS -- (index, the 11 dense floats, dRGB): 64 bytes per Gaussian
        # with a gradient -- instead of summing 44 bytes and gathering 12 per Gaussian: counts first (one small all-gather
        # the host waits for: it sizes the padded row all-gather), the rows, then every rank adds the views' rows into a
        # zeroed dense part in VIEW ORDER (one scatter launch per view) and rebuilds the SH sum as the factored plan does.
        # Bit-identical to the single-process accumulation, whatever the collective's internals.
        self.sparse = bool(sparse) and "shs" in self.shapes
        factor_sh = factor_sh or self.sparse
        self.group = process_group
        self.average = average
        _, self.layout, self.numel = flat_layout(self.shapes)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.views = {k: self.flat[o:o + n].view(self.shapes[k]) for k, (o, n) in self.layout.items()}
        self.factor_sh = bool(factor_sh) and "shs" in self.shapes
        if reduce not in ("allreduce", "direct"):
            raise ValueError("reduce must be 'allreduce' or 'direct'")
        self.reduce = reduce
        self._direct = None          # (send view, recv buffer, shard length) of the direct plan, sized at first start()
        self._works = []
        if self.factor_sh:
            # "shs" is last in PARAM_ORDER: the dense prefix is everything before it
            self.dense = self.flat[: self.layout["shs"][0]]
            P = self.shapes["shs"][0]
            self.payload_numel = 3 * P + 4                  # dRGB[P,3], camera centre[3], pad
            self.own = torch.zeros(self.payload_numel, dtype=torch.float32, device=self.device)
            self.own_drgb = self.own[: 3 * P].view(P, 3)
            self.own_campos = self.own[3 * P: 3 * P + 3]
            self.gathered = None                            # [world, payload_numel], sized at first start()
            self.sh_reducer = sh_reducer or _hip_sh_reducer
            self.means3D, self.sh_degree = None, 0
        if self.sparse:
            self.P = P
            self.row_packer = row_packer or _hip_row_packer
            self.row_scatterer = row_scatterer or _hip_row_scatterer
            # room for every Gaussian's row (64 B each: 192 MB at 3 M -- 288 GB of HBM per GPU): a view can never overflow it,
            # and only the padded prefix the gathered counts call for travels
            self.rows_own = torch.zeros((P + 256, ROW_FLOATS), dtype=torch.float32, device=self.device)
            self.header_own = torch.zeros(4, dtype=torch.float32, device=self.device)     # row count (int32 bit pattern), camera centre
            self.count_dev = self.header_own[0:1].view(torch.int32)                       # (the pack kernel writes the count here)
            self.own_campos = self.header_own[1:4]                                        # (the backward's payload step writes the centre here)
            self.headers = None                        # [world, 4]
            self.rows_all = None                       # [world, capacity, ROW_FLOATS], sized from the gathered counts
            self.counts = []                           # rows per view of the exchange in flight
            self.sparse_stats = {"rows_own": 0, "rows_max": 0}

    def set_sh_context(self, means3D: torch.Tensor, sh_degree: int):
        """Replicated inputs the SH rebuild needs (factored plan)."""
        self.means3D, self.sh_degree = means3D, int(sh_degree)

    @property
    def nbytes(self) -> int:
        return self.numel * 4

    @property
    def wire_floats_per_rank(self) -> int:
        """Floats each rank contributes to the collectives of one exchange."""
        if self.sparse:       # what the last exchange moved: the padded rows + the header
            return max(self.sparse_stats["rows_max"], 1) * ROW_FLOATS + 4
        return (self.dense.numel() + self.payload_numel) if self.factor_sh else self.numel

    def _active(self):
        import torch.distributed as dist
        return self.group is not None and dist.is_initialized()

    def all_reduce(self):
        """Blocking (stream-ordered) SUM over the group; returns the flat buffer."""
        self.start()
        return self.wait()

    def start(self, part: str = "all"):
        """Enqueue the collectives behind the work already on the current stream and return at
        once (torch.distributed async_op): they run on the backend's own stream, so kernels
        enqueued afterwards on the compute stream overlap with them.
        part (factored plan): "gather" = only the all-gather of the colour-gradient payloads (they are complete after
        phase 1 of a two-call backward), "dense" = only the sum of the dense part (after phase 2); "all" = both."""
        import torch.distributed as dist
        if part != "dense":
            self._works = []
        if not self._active():
            return None
        if self.sparse:
            return self._start_sparse()
        if self.factor_sh:
            if part != "dense":
                world = dist.get_world_size(self.group)
                if self.gathered is None or self.gathered.shape[0] != world:
                    self.gathered = torch.zeros((world, self.payload_numel), dtype=torch.float32, device=self.device)
                self._works.append(dist.all_gather_into_tensor(self.gathered.view(-1), self.own, group=self.group, async_op=True))
            if part == "gather":
                return self._works
            summed = self.dense
        else:
            summed = self.flat
        if self.reduce == "direct":
            self._start_direct(summed)
        else:
            self._works.append(dist.all_reduce(summed, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return self._works

    def _start_sparse(self):
        """Pack this view's non-zero rows (the kernel leaves their number in the header), all-gather the headers -- row
        count + camera centre of every view -- and read them: the ONE host wait of the plan (the counts size the padded
        all-gather of the rows; the zero fills the scatter needs are enqueued before it and run while the host waits).
        Then the all-gather of the rows is enqueued."""
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        if self.headers is None or self.headers.shape[0] != world:
            self.headers = torch.zeros((world, 4), dtype=torch.float32, device=self.device)
        if self.gathered is None or self.gathered.shape[0] != world:
            self.gathered = torch.zeros((world, self.payload_numel), dtype=torch.float32, device=self.device)
        self.row_packer(self)                                    # rows_own, header_own[0] <- the count
        dist.all_gather_into_tensor(self.headers.view(-1), self.header_own, group=self.group)
        self.dense.zero_()                                       # behind the pack (it reads the dense part), under the host's wait
        self.gathered.zero_()
        hdr = self.headers.cpu()                                 # host wait: backward, pack and the header gather are through
        self.counts = [int(c) for c in hdr[:, 0].contiguous().view(torch.int32).tolist()]
        cap = (max(self.counts + [1]) + 255) // 256 * 256      # the same on every rank: it comes from the gathered counts
        if self.rows_all is None or self.rows_all.shape[0] != world or self.rows_all.shape[1] < cap:
            self.rows_all = torch.zeros((world, int(cap * 1.25) // 256 * 256 + 256, ROW_FLOATS), dtype=torch.float32, device=self.device)
        self._rows_cap = cap
        self._rows_recv = self.rows_all.view(-1)[: world * cap * ROW_FLOATS]
        self.sparse_stats.update(rows_own=self.counts[rank], rows_max=max(self.counts))
        self._works.append(dist.all_gather_into_tensor(self._rows_recv, self.rows_own[:cap].reshape(-1), group=self.group, async_op=True))
        return self._works

    def _finish_sparse(self):
        """The dense part <- the views' rows added in view order into zeros; the dRGB of every view laid out densely for
        the SH rebuild; the rebuild."""
        world, cap = len(self.counts), self._rows_cap
        P = self.P
        recv = self._rows_recv.view(world, cap, ROW_FLOATS)
        self.gathered[:, 3 * P: 3 * P + 3].copy_(self.headers[:, 1:4])
        for v in range(world):                                # view order: the order of the single-process accumulation
            if self.counts[v]:
                self.row_scatterer(self, recv[v], self.counts[v], self.gathered[v, : 3 * P])
        if self.means3D is None:
            raise RuntimeError("sparse exchange: call set_sh_context(means3D, sh_degree) first")
        self.sh_reducer(self)

    def _start_direct(self, summed):
        """Phase 1 of the direct plan.  RCCL: ONE reduce_scatter_tensor -- rank j receives the sum of everyone's shard
        j.  Backends without it (gloo, the CPU tests): an all-to-all of the N shards and a local sum in rank order.
        The tail shard is zero-padded in a staging copy only when the length does not divide.  Phase 2 (all-gather of
        the reduced shards) runs in wait()."""
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        self._direct_rs = dist.get_backend(self.group) == "nccl"
        n = summed.numel()
        shard = (n + world - 1) // world
        if self._direct is None or self._direct[2] != shard or self._direct[1].shape[0] != world:
            send = summed if shard * world == n else torch.zeros(shard * world, dtype=torch.float32, device=self.device)
            recv = torch.empty((world, shard), dtype=torch.float32, device=self.device)
            mine = torch.empty(shard, dtype=torch.float32, device=self.device)
            self._direct = (send, recv, shard, mine)
        send, recv, shard, mine = self._direct
        if send is not summed:
            send[:n].copy_(summed)
        self._direct_target = summed
        if self._direct_rs:
            self._works.append(dist.reduce_scatter_tensor(mine, send, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_to_all_single(recv.view(-1), send, group=self.group, async_op=True))

    def _finish_direct(self):
        import torch.distributed as dist
        send, recv, shard, mine = self._direct
        summed = self._direct_target
        n = summed.numel()
        if not self._direct_rs:
            torch.sum(recv, dim=0, out=mine)      # fixed order over source ranks: identical on every rank after the gather
        world = recv.shape[0]
        if shard * world == n:
            dist.all_gather_into_tensor(summed, mine, group=self.group)
        else:
            dist.all_gather_into_tensor(send, mine, group=self.group)
            summed.copy_(send[:n])

    def wait(self):
        """Make the current stream (CPU backends: the caller) wait for the pending collectives;
        in the factored plan, then rebuild the summed SH gradient on the current stream."""
        self.join()
        if self._works:
            self._finish_on_current_stream()
        return self.flat

    def _finish_on_current_stream(self):
        import torch.distributed as dist
        for w in self._works:
            w.wait()
        self._works = []
        if self.sparse:
            self._finish_sparse()
        elif self.reduce == "direct":
            self._finish_direct()
        if self.factor_sh and not self.sparse:
            if self.means3D is None:
                raise RuntimeError("factored exchange: call set_sh_context(means3D, sh_degree) first")
            self.sh_reducer(self)
        if self.average:
            self.flat.mul_(1.0 / dist.get_world_size(self.group))

    def finish_in_step(self):
        """The schedule of a training step that updates the parameters before the next view is rendered: everything
        started by start() is complete (on the current stream) when this returns -- nothing is left for a later step.
        Inside it, factored plan on a GPU: the all-gather (started first, 3 floats per Gaussian) is waited for alone
        and the SH rebuild it feeds runs on a side stream WHILE the sum of the dense part is still on the wire; the
        current stream then waits for both."""
        if not self._works:
            return self.flat
        overlap = self.factor_sh and not self.sparse and not self.average and self.flat.device.type == "cuda"
        if not overlap:
            self._finish_on_current_stream()
            return self.flat
        if self.means3D is None:
            raise RuntimeError("factored exchange: call set_sh_context(means3D, sh_degree) first")
        self._works[0].wait()                          # the all-gather of dRGB / camera centres
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(self.flat.device)
        cur = torch.cuda.current_stream(self.flat.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            self.sh_reducer(self)                      # writes views["shs"] only: disjoint from the dense part
        for w in self._works[1:]:
            w.wait()
        self._works = []
        if self.reduce == "direct":
            self._finish_direct()
        cur.wait_stream(self._side)
        return self.flat

    def wait_on_side_stream(self):
        """GPU, factored plan: the current stream waits for the pending collectives (so the dense part
        may be overwritten by the next backward), but the SH rebuild -- an HBM-bound 0.15 ms kernel that
        only touches the SH rows and the gathered colour gradients -- goes to a side stream ordered after
        what the caller has enqueued so far, and overlaps what it enqueues next (the backward blend is
        VALU bound).  join() / wait() orders the caller's stream after it: call it before the exchange of
        this buffer is started again and before the SH rows are read."""
        if not self._works:
            return
        if not self.factor_sh or self.sparse or self.average or self.flat.device.type != "cuda":
            self._finish_on_current_stream()
            return
        if self.means3D is None:
            raise RuntimeError("factored exchange: call set_sh_context(means3D, sh_degree) first")
        for w in self._works:
            w.wait()
        self._works = []
        if self.reduce == "direct":
            self._finish_direct()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(self.flat.device)
        self._side.wait_stream(torch.cuda.current_stream(self.flat.device))
        with torch.cuda.stream(self._side):
            self.sh_reducer(self)
            self._joined = torch.cuda.Event()
            self._joined.record(self._side)

    def join(self):
        ev = getattr(self, "_joined", None)
        if ev is not None:
            torch.cuda.current_stream(self.flat.device).wait_event(ev)
            self._joined = None


# ---- slot-sum exchange (round 6) ---------------------------------------------------------------------------------------------
SUM_HDR_WORDS = 64      # frosting_amd/csrc/slot_exchange.hip: the packet's header ...
SUM_ROW_FLOATS = 12     # ... and its rows {masked dRGB[3], six pixel moments, three view-direction terms}
SUM_TILE = 1536         # chunk boundaries fall on whole tiles of the combine pass, whichever tile size it picks (3 .. 24 blocks of 64)


def sum_packet_words(n: int, capacity: int) -> int:
    """32-bit words of a packet of n Gaussians with room for `capacity` rows (= frg_sum_packet_bytes / 4): header, one bit per
    Gaussian, one row offset per block of 64, the rows."""
    nblk = (n + 63) // 64
    rows_at = (SUM_HDR_WORDS + 2 * nblk + nblk + 3) // 4 * 4
    return (rows_at + SUM_ROW_FLOATS * capacity + 3) // 4 * 4


def _hip_sum_packer(ex: "SlotSumExchange", chunk: int, dest: torch.Tensor):
    """dest <- the packet of chunk `chunk` of THIS rank's view (frg_pack_sum_rows on the workspace phase 1 of the backward left)."""
    if dest.device.type != "cuda":
        raise RuntimeError("the slot-sum exchange packs its rows with a HIP kernel (no CPU path; tests inject their own packer)")
    c = ex.view_ctx
    first, count = ex.chunks[chunk]
    cam = c["cam"]
    stream = C.c_void_p(torch.cuda.current_stream(dest.device).cuda_stream)
    rc = _lib.lib().frg_pack_sum_rows(ex.P, int(c["R"]), first, count, _p(c["work"]), c["work"].numel(), _p(ex.own_drgb),
                                      _p(cam.viewmatrix), _p(cam.projmatrix), _p(cam.campos), float(cam.tanfovx), float(cam.tanfovy),
                                      int(cam.image_width), int(cam.image_height), float(c.get("scale_modifier", 1.0)), int(c["D"]),
                                      _p(dest), dest.numel() * 4, int(ex.capacity[chunk]), stream)
    if rc < 0:
        raise RuntimeError(f"frg_pack_sum_rows failed ({rc}): {_lib.last_error()}")


def _hip_sum_combiner(ex: "SlotSumExchange", chunk: int, packets: torch.Tensor, n_views: int, seq: int):
    """The flat gradient buffer's rows of chunk `chunk` <- the sum over the views' packets, in view order (frg_backward_combine)."""
    if packets.device.type != "cuda":
        raise RuntimeError("the slot-sum exchange combines the views with a HIP kernel (no CPU path; tests inject their own)")
    first, count = ex.chunks[chunk]
    g, pr = ex.views, ex.params
    raw = ex.raw_params
    v = lambda t: None if t is None else t.data_ptr()
    ws = int(_lib.lib().frg_combine_workspace_bytes(int(n_views), int(ex.capacity[chunk])))
    if ex.combine_work is None or ex.combine_work.numel() < ws:
        ex.combine_work = torch.empty(int(ws * 1.25) + 256, dtype=torch.uint8, device=packets.device)
    a = _lib.CombineArgs(struct_size=C.sizeof(_lib.CombineArgs), P=ex.P, first=first, count=count, n_views=int(n_views),
                         packets=v(packets), packet_stride_bytes=packets.shape[1] * 4, capacity_rows=int(ex.capacity[chunk]),
                         M=int(ex.shapes["shs"][1]), means3D=v(pr["means3D"]), shs=v(pr["shs"]),
                         scales=None if raw else v(pr["scales"]), rotations=None if raw else v(pr["rotations"]),
                         opacities=None if raw else v(pr["opacities"]), raw_opacities=v(pr["opacities"]) if raw else None,
                         raw_scales=v(pr["scales"]) if raw else None, raw_rotations=v(pr["rotations"]) if raw else None,
                         dL_dmean3D=v(g["means3D"]), dL_dscale=v(g["scales"]), dL_drot=v(g["rotations"]), dL_dopacity=v(g["opacities"]),
                         dL_dsh=v(g["shs"]), status=v(ex.status[chunk]), status_seq=int(seq), row_live=v(getattr(ex, "row_live", None)),
                         workspace=v(ex.combine_work), workspace_bytes=ex.combine_work.numel(), hip_stream=torch.cuda.current_stream(packets.device).cuda_stream)
    rc = _lib.lib().frg_backward_combine(C.byref(a))
    if rc < 0:
        raise RuntimeError(f"frg_backward_combine failed ({rc}): {_lib.last_error()}")


class SlotSumExchange(GradientExchange):
    """The exchange plan of round 6: ranks all-gather the nine per-Gaussian SLOT SUMS of their view's backward (phase 1) for
    the Gaussians that have any -- 48-byte rows {masked dRGB, six moments, three view-direction terms} in index order behind a bit mask, a fixed-capacity packet per chunk of
    Gaussians -- and every rank runs the per-Gaussian chain (phase 2) for EVERY view's rows itself, in view order, in one pass
    that writes each of the 59 gradient floats of a Gaussian once (csrc/slot_exchange.hip).  Bit-identical to accumulating the
    per-view gradients in one process.  Against the factored plan: a quarter of the wire bytes, no dense zero fills, no
    per-view scatters, no SH rebuild, and no host wait for a count in front of the collective:

      * capacity: the packets are sized from what the PREVIOUS step's views wanted (x slack), the first step for every
        Gaussian; the wanted counts arrive with the packets and the combine pass posts them to pinned host memory as it
        starts -- the host learns the verdict while the pass runs.  A view that wants more than the capacity makes every
        rank (they all see the same headers) pack, gather and combine that chunk again with a larger one: rare, and correct.
      * chunks: the Gaussians are cut into `chunks` index ranges with a packet and a collective each, so that the combine pass
        of one range runs while the next range's packets are still on the wire.

    Rank-local outputs of the backward (dL_dmeans2D -- the viewspace gradient --, dL_dcov3D, dL_dcolors) are NOT produced on
    this path: phase 2 never runs for the own view alone."""

    slotsum = True

    def __init__(self, shapes: dict, device, process_group=None, average: bool = False, chunks: int = 2, slack: float = 1.125,
                 packer=None, combiner=None, raw_params: bool = False):
        super().__init__(shapes, device, process_group, average)
        if "shs" not in self.shapes:
            raise ValueError("the slot-sum exchange needs the SH parameterisation (shs)")
        P = self.shapes["means3D"][0]
        self.P = P
        self.raw_params = raw_params
        self.own_drgb = torch.zeros((P, 3), dtype=torch.float32, device=self.device)   # clamp-masked colour gradient, by phase 1
        # chunk boundaries on whole tiles of the combine pass
        tiles = (P + SUM_TILE - 1) // SUM_TILE
        k = max(1, min(int(chunks), tiles))
        cuts = [min(P, (tiles * i // k) * SUM_TILE) for i in range(k)] + [P]
        self.chunks = [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(k) if cuts[i + 1] > cuts[i]]
        self.capacity = [n for _, n in self.chunks]                # first step: room for every Gaussian's row
        self.slack = float(slack)
        self.packer = packer or _hip_sum_packer
        self.combiner = combiner or _hip_sum_combiner
        pin = self.device.type == "cuda"
        # the combine pass's verdict per chunk: 64-bit words (sequence number << 32 | value) -- [0] overflow, [1 + v] rows view v wanted
        self.status = [torch.zeros(1 + 16, dtype=torch.int64, pin_memory=pin) for _ in self.chunks]
        self.seq = 0
        self.packet_own = [None] * len(self.chunks)
        self.packets_all = [None] * len(self.chunks)
        self.view_ctx = None
        self.params = None
        self.stats = {"rows_wanted_max": 0, "rows_per_view_max": 0, "repacks": 0, "packet_bytes": 0}
        self._no_post = False
        # optional uint8 [P]: the combine pass then marks the Gaussians that have a row in some view and does NOT write the rows of the
        # others (frg_combine_args::row_live) -- for a consumer that takes an unmarked row as zero (FlatAdam.step(row_live=...))
        self.row_live = None
        self.combine_work = None      # the combine pass's scratch (256 bytes since the pass is one kernel; the ABI keeps the argument)

    def set_params(self, params: dict):
        """The replicated parameters the combine pass reads: {'means3D','shs','scales','rotations','opacities'} (raw forms
        with raw_params)."""
        self.params = params

    def note_view(self, **ctx):
        """What the packer needs of the backward that just ran its phase 1: P-sized workspace `work`, R, cam, D, scale_modifier."""
        self.view_ctx = ctx

    @property
    def wire_floats_per_rank(self) -> int:
        return sum(sum_packet_words(n, cap) for (_, n), cap in zip(self.chunks, self.capacity))

    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group)

    def _ensure(self, c: int, world: int):
        words = sum_packet_words(self.chunks[c][1], self.capacity[c])
        if self.packet_own[c] is None or self.packet_own[c].numel() != words:
            self.packet_own[c] = torch.zeros(words, dtype=torch.int32, device=self.device)
        if self.packets_all[c] is None or tuple(self.packets_all[c].shape) != (world, words):
            self.packets_all[c] = torch.zeros((world, words), dtype=torch.int32, device=self.device)

    def _gather(self, c: int, world: int, async_op: bool):
        import torch.distributed as dist
        self._ensure(c, world)
        self.packer(self, c, self.packet_own[c])
        return dist.all_gather_into_tensor(self.packets_all[c].view(-1), self.packet_own[c], group=self.group, async_op=async_op)

    def start(self, part: str = "all"):
        """Pack this view's chunks and enqueue their all-gathers (one per chunk, in order) behind the work on the current stream."""
        self._works = []
        if not self._active():
            return None
        if self.view_ctx is None:
            raise RuntimeError("slot-sum exchange: note_view() after the backward's phase 1 first")
        world = self._world()
        self._works = [self._gather(c, world, True) for c in range(len(self.chunks))]
        self.stats["packet_bytes"] = 4 * self.wire_floats_per_rank
        return self._works

    def start_chunk(self, c: int):
        """Pack chunk c of this view and enqueue its all-gather behind the work on the current stream -- for a backward whose
        phase 1 comes in pieces (ViewParallelRasterizer.backward_overlapped): in chunk order, _works cleared by the caller first."""
        if not self._active():
            return None
        if self.view_ctx is None:
            raise RuntimeError("slot-sum exchange: note_view() after the backward's phase 1 first")
        if len(self._works) != c:
            raise RuntimeError(f"slot-sum exchange: chunk {c} started out of order ({len(self._works)} in flight)")
        self._works.append(self._gather(c, self._world(), True))
        self.stats["packet_bytes"] = 4 * self.wire_floats_per_rank
        return self._works[-1]

    def _finish_on_current_stream(self):
        import torch.distributed as dist
        if self.params is None:
            raise RuntimeError("slot-sum exchange: call set_params() first")
        world = self._world()
        self.seq = self.seq % 0x7fffffff + 1
        for c, w in enumerate(self._works):                       # chunk c's combine pass runs while chunk c + 1 travels
            w.wait()
            self.combiner(self, c, self.packets_all[c], world, self.seq)
        self._works = []
        wanted = [self._verdict(c, world) for c in range(len(self.chunks))]
        for c, (over, counts) in enumerate(wanted):
            if over:         # every rank reads the same headers: the same decision everywhere, no collective about it
                intact = (self.view_ctx or {}).get("intact")
                if intact is not None and not intact():
                    raise RuntimeError("slot-sum exchange: a view wanted more rows than its packet holds, and the backward's workspace has "
                                       "been overwritten since (the stale-overlap schedule): use the in-step schedule")
                self.capacity[c] = min(self.chunks[c][1], (int(max(counts) * self.slack) // 256 + 1) * 256)
                self.stats["repacks"] += 1
                self._gather(c, world, False)
                self.seq = self.seq % 0x7fffffff + 1
                self.combiner(self, c, self.packets_all[c], world, self.seq)
                over2, counts = self._verdict(c, world)
                if over2:
                    raise RuntimeError(f"slot-sum exchange: chunk {c} still overflows a capacity of {self.capacity[c]} rows (wanted {max(counts)})")
            else:            # the next step's packets: what this step's views wanted, with slack (shrinks only by a clear margin)
                want = min(self.chunks[c][1], (int(max(counts) * self.slack) // 256 + 1) * 256)
                if want < 0.8 * self.capacity[c] or want > self.capacity[c]:
                    self.capacity[c] = want
            self.stats["rows_wanted_max"] = max(self.stats["rows_wanted_max"], max(counts))
            wanted[c] = (False, counts)
        # rows per view over ALL chunks of this step (the largest view's): what the packets' capacity follows
        self.stats["rows_per_view_max"] = max(sum(cs[v] for _, cs in wanted) for v in range(world))
        if self.average:
            self.flat.mul_(1.0 / dist.get_world_size(self.group))

    def _verdict(self, c: int, world: int):
        """(overflow?, rows wanted per view) of chunk c's last combine pass: posted by the pass to pinned host memory as it
        starts (GPU), so the host polls instead of synchronising; written by the stand-in directly (CPU tests)."""
        st = self.status[c]
        words = st[:1 + world]

        def posted():
            w = words.tolist()
            return all((x >> 32) == self.seq for x in w), w
        if self.device.type == "cuda":
            stream = torch.cuda.current_stream(self.device)
            spins, (ok, w) = 0, posted()
            while not self._no_post and not ok:
                spins += 1
                if spins % 64 == 0 and stream.query():
                    ok, w = posted()
                    if not ok:
                        self._no_post = True    # the stream drained without the post becoming visible: read the headers instead, from now on
                    break
                ok, w = posted()
            if self._no_post:
                hdr = self.packets_all[c][:world, :6].cpu()        # (synchronises: the plain way)
                wants, caps = [int(x) for x in hdr[:, 1].tolist()], [int(x) for x in hdr[:, 3].tolist()]
                return any(a > k for a, k in zip(wants, caps)), wants
        else:
            ok, w = posted()
            if not ok:
                raise RuntimeError("slot-sum exchange: the combine stand-in posted no verdict")
        return bool(w[0] & 0xffffffff), [int(x & 0xffffffff) for x in w[1:]]

    def finish_in_step(self):
        if not self._works:
            return self.flat
        self._finish_on_current_stream()
        return self.flat

    # ---- one process playing every rank (tests, tools/combine_bench.py): view after view, then one combine ----
    def pack_local_view(self, view_index: int, world: int):
        """The packets of the view whose phase 1 just ran go where an all-gather would have put rank `view_index`'s."""
        for c in range(len(self.chunks)):
            self._ensure(c, world)
            self.packer(self, c, self.packets_all[c][view_index])

    def combine_local(self, world: int):
        """The combine pass over `world` locally packed views -> [(overflow?, rows wanted per view)] per chunk."""
        self.seq = self.seq % 0x7fffffff + 1
        for c in range(len(self.chunks)):
            self.combiner(self, c, self.packets_all[c], world, self.seq)
        return [self._verdict(c, world) for c in range(len(self.chunks))]

    def wait_on_side_stream(self):
        if self._works:
            self._finish_on_current_stream()


def probe_reduce_plan(exchanges, iters: int = 5, warm: int = 3):
    """Time the sum of the dense part both ways -- the backend's all-reduce, and reduce-scatter + all-gather of 1/N shards
    ("direct") -- on the exchange buffers as they are (contents are summed over and over: garbage in, garbage out; the
    caller zeroes or overwrites them afterwards) and set every exchange to the faster plan.  Collective: every rank of the
    group must call it.  Returns ({"allreduce": ms, "direct": ms} with the MAX over ranks, the plan chosen) -- the same on
    every rank, because the choice is made from the reduced times."""
    import time
    import torch.distributed as dist
    ex = exchanges[0]
    dev = ex.flat.device
    times = {}
    for mode in ("allreduce", "direct"):
        for e in exchanges:
            e.reduce = mode
        t0 = 0.0
        for it in range(warm + iters):
            if it == warm:
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
                dist.barrier(group=ex.group)
                t0 = time.perf_counter()
            ex.start(part="dense")
            for w in ex._works:
                w.wait()
            ex._works = []
            if mode == "direct":
                ex._finish_direct()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        tm = torch.tensor([(time.perf_counter() - t0) / iters], dtype=torch.float64, device=dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX, group=ex.group)
        times[mode] = 1e3 * float(tm.item())
    best = min(times, key=times.get)
    for e in exchanges:
        e.reduce = best
    return times, best


class DensificationStats:
    """The per-Gaussian statistics vanilla 3DGS densification keeps (gaussian_splatting/train.py:116-117,
    gaussian_splatting/scene/gaussian_model.py:404-407), over a view-parallel batch: every rank holds one view's
    `radii` and screen-space gradient; the batch's contribution is
        max_radii2D         = max(max_radii2D, max over views of radii)          on the Gaussians visible in that view
        xyz_gradient_accum += sum over visible views of ||viewspace_grad[:, :2]||
        denom              += number of views in which the Gaussian was visible
    i.e. ONE all-reduce(MAX) of P int32 and ONE all-reduce(SUM) of 2 P floats.  Frosting's
    refinement does not densify (refine.py has no such step), so this is optional on the C5 path."""

    def __init__(self, P: int, device, process_group=None):
        self.group = process_group
        dev = torch.device(device)
        self.max_radii2D = torch.zeros(P, dtype=torch.float32, device=dev)
        self.xyz_gradient_accum = torch.zeros(P, 1, dtype=torch.float32, device=dev)
        self.denom = torch.zeros(P, 1, dtype=torch.float32, device=dev)
        self._radii = torch.zeros(P, dtype=torch.int32, device=dev)
        self._sums = torch.zeros(P, 2, dtype=torch.float32, device=dev)

    def update(self, radii: torch.Tensor, viewspace_grad: torch.Tensor):
        """radii [P] int32, viewspace_grad [P,3] (dL_dmeans2D) of THIS rank's view; collective over the group."""
        import torch.distributed as dist
        vis = radii > 0
        torch.where(vis, radii, torch.zeros_like(radii), out=self._radii)
        self._sums[:, 0] = torch.where(vis, viewspace_grad[:, :2].norm(dim=-1), torch.zeros_like(self._sums[:, 0]))
        self._sums[:, 1] = vis.to(torch.float32)
        if self.group is not None and dist.is_initialized():
            works = [dist.all_reduce(self._radii, op=dist.ReduceOp.MAX, group=self.group, async_op=True),
                     dist.all_reduce(self._sums, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
            for w in works:
                w.wait()
        torch.maximum(self.max_radii2D, self._radii.to(torch.float32), out=self.max_radii2D)
        self.xyz_gradient_accum += self._sums[:, 0:1]
        self.denom += self._sums[:, 1:2]


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

    def __init__(self, scene, device, process_group=None, average: bool = False, factor_sh: bool = False,
                 deferred_counters: bool = False, capacity_slack: float = 1.25, reduce: str = "allreduce",
                 write_all_outputs: bool = True, raw_params: bool = False, sparse: bool = False, live_rows: bool = False,
                 slotsum: bool = False, chunks: int = 2, phase1_in_pieces: bool = False):
        """deferred_counters: after the first (synchronous) view, forwards run through
        frg_forward_deferred -- no host synchronisation inside the step; finish() then reports the
        true instance count and whether the view has to be repeated (capacity exceeded)."""
        self.dev = torch.device(device)
        # raw_params: scene.opacities / scales / rotations hold the model's RAW parameters (logit, log, unnormalised
        # quaternion); the activations run inside the per-Gaussian kernels (frg_forward_ex / frg_backward_ex) and the
        # gradients written to the flat buffer are those of the raw parameters -- straight into the optimizer
        self.raw_params = raw_params
        # live_rows: the backward marks the Gaussians with a gradient in self.row_live (uint8 [P]) and does NOT write the
        # rows of the others (frg_backward_args::row_live) -- for a consumer that takes an unmarked row as zero without
        # reading it (FlatAdam.step(row_live=...)).  Single-GPU training steps; not with an exchange (it sums whole buffers).
        self.live_rows = live_rows
        # slot-sum plan, several chunks: the backward's phase 1 runs chunk by chunk and every chunk's packet leaves as soon as
        # its sums exist (backward_overlapped): the wire starts ~0.1 ms earlier at C3
        self.phase1_in_pieces = phase1_in_pieces
        self.deferred_counters = deferred_counters
        self.capacity_slack = capacity_slack
        self.capacity = 0            # instances the binning arena is sized for (deferred forwards)
        self._pending = False
        self.scene = scene
        P, K = scene.means3D.shape[0], scene.shs.shape[1]
        self.P, self.K = P, K
        self._group, self._average = process_group, average
        shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
        # two gradient buffers: the exchange of step k may still be in flight on the
        # collective stream while step k+1 renders and writes the other buffer
        # sparse: the exchange moves rows of the Gaussians with a gradient (GradientExchange, third plan)
        # slotsum: the ranks exchange the slot sums of phase 1 and every rank runs phase 2 for every view (SlotSumExchange, round 6)
        self._chunks = chunks
        self.exchanges = [self._make_exchange(shapes, "slotsum" if slotsum else "sparse" if sparse else "factored" if factor_sh else "allreduce", reduce)
                          for _ in range(2)]
        self.exchange = self.exchanges[0]
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=self.dev)
        # rank-local (not exchanged) backward outputs.  dL_dcov3D is an intermediate of the chain when the covariance
        # comes from scales / rotations, but it is one of the eight tensors the reference's backward returns
        # (rasterize_points.cu:195) and part of SURVEY 8(d)'s byte model (24 of the 284 B per Gaussian): written unless
        # the caller opts out (write_all_outputs=False)
        self.dL_dmeans2D, self.dL_dcolors = f(P, 3), f(P, 3)
        self.write_all_outputs = write_all_outputs
        self.dL_dcov3D = f(P, 6) if write_all_outputs else None
        self.geom, self.binning, self.img, self.work = (_Arena(self.dev) for _ in range(4))
        self.radii = torch.empty(P, dtype=torch.int32, device=self.dev)
        self.row_live = torch.zeros(P, dtype=torch.uint8, device=self.dev) if live_rows else None
        if live_rows and process_group is not None:
            raise ValueError("live_rows leaves the rows of Gaussians without a gradient unwritten: not with a gradient exchange")
        self.out_color = None
        self.num_rendered = 0
        self._view = None

    def _make_exchange(self, shapes, plan: str, reduce: str):
        s = self.scene
        if plan == "slotsum":
            ex = SlotSumExchange(shapes, self.dev, self._group, self._average, chunks=self._chunks, raw_params=self.raw_params)
            ex.set_params(dict(means3D=s.means3D, shs=s.shs, scales=s.scales, rotations=s.rotations, opacities=s.opacities))
            return ex
        ex = GradientExchange(shapes, self.dev, self._group, self._average, factor_sh=(plan != "allreduce"), reduce=reduce,
                              sparse=(plan == "sparse"))
        if ex.factor_sh:
            ex.set_sh_context(s.means3D, s.sh_degree)
        return ex

    def set_exchange_plan(self, plan: str, reduce: str = None):
        """Replace the gradient exchange: 'allreduce' | 'factored' | 'sparse' (GradientExchange) | 'slotsum' (SlotSumExchange).
        Pending collectives of the old plan must have been waited for; the gradient buffers are new (zeroed)."""
        old = self.exchanges[0]
        shapes = old.shapes
        reduce = reduce or old.reduce
        self.exchanges = None
        del old
        self.exchanges = [self._make_exchange(shapes, plan, reduce) for _ in range(2)]
        self.exchange = self.exchanges[0]

    def forward(self, cam, bg, deferred=None, keep_mask=None, forward_only=False):
        """Render one view.  With deferred counters the returned image is valid only if the
        following finish() returns True.  keep_mask (bool / uint8 [P], optional): Frosting's occlusion
        culling as a skip flag (frg_forward_ex).  forward_only: no backward() will follow this view
        (frg_forward_args::forward_only; not with deferred counters)."""
        L = _lib.lib()
        s = self.scene
        H, W = cam.image_height, cam.image_width
        if self.out_color is None or tuple(self.out_color.shape) != (3, H, W):
            self.out_color = torch.empty((3, H, W), dtype=torch.float32, device=self.dev)
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        # the blend arithmetic of this forward (the process option, read here) is handed to the backward explicitly: the library
        # then launches that one instantiation instead of both behind the forward's stamp (api.hip backward_impl)
        self._exact = int(_lib.get_option("exact_blend"))
        use_deferred = (self.deferred_counters if deferred is None else deferred) and self.capacity > 0
        if forward_only and use_deferred:
            raise RuntimeError("forward_only is not offered with deferred counters")
        if keep_mask is not None or self.raw_params or forward_only:
            v = lambda t: None if t is None else t.data_ptr()
            rawkw = {}
            if self.raw_params:
                rawkw = dict(raw_opacities=v(s.opacities), raw_scales=v(s.scales), raw_rotations=v(s.rotations))
            a = _lib.ForwardArgs(
                struct_size=C.sizeof(_lib.ForwardArgs), geometry_alloc=self.geom.cb, binning_alloc=self.binning.cb,
                image_alloc=self.img.cb, user=None, P=self.P, D=s.sh_degree, M=self.K, background=v(bg), width=W, height=H,
                means3D=v(s.means3D), shs=v(s.shs), colors_precomp=None, opacities=None if self.raw_params else v(s.opacities),
                scales=None if self.raw_params else v(s.scales),
                scale_modifier=1.0, rotations=None if self.raw_params else v(s.rotations), cov3D_precomp=None, viewmatrix=v(cam.viewmatrix),
                projmatrix=v(cam.projmatrix), cam_pos=v(cam.campos), tan_fovx=float(cam.tanfovx), tan_fovy=float(cam.tanfovy),
                prefiltered=0, out_color=v(self.out_color), radii=v(self.radii), debug=0, hip_stream=stream.value,
                instance_capacity=self.capacity if use_deferred else 0, keep_mask=v(keep_mask),
                forward_only=1 if forward_only else 0, **rawkw)
            self._keep_alive = keep_mask
            rc = L.frg_forward_ex(C.byref(a))
        else:
            fn = L.frg_forward_deferred if use_deferred else L.frg_forward
            rc = fn(self.geom.cb, self.binning.cb, self.img.cb, None,
                    self.P, s.sh_degree, self.K, _p(bg), W, H,
                    _p(s.means3D), _p(s.shs), None, _p(s.opacities),
                    _p(s.scales), 1.0, _p(s.rotations), None,
                    _p(cam.viewmatrix), _p(cam.projmatrix), _p(cam.campos),
                    float(cam.tanfovx), float(cam.tanfovy), 0,
                    _p(self.out_color), _p(self.radii), self.capacity if use_deferred else 0, stream)
        if rc < 0:
            raise RuntimeError(f"{'frg_forward_deferred' if use_deferred else 'frg_forward'} failed ({rc}): {_lib.last_error()}")
        # deferred: rc is the capacity, which is what the backward carves its buffers with
        self.num_rendered = rc
        self._pending = use_deferred and rc > 0
        if not use_deferred:
            self.true_num_rendered = rc
            if self.deferred_counters:
                self.capacity = max(self.capacity, int(rc * self.capacity_slack) + 4096)
        self._view = (cam, bg)
        return self.out_color, self.radii

    def finish(self) -> bool:
        """Deferred counters: wait for the last forward's counters (not for its kernels).  True: the
        view was rasterized (true_num_rendered is set).  False: it had more instances than the
        capacity -- nothing was rasterized; the capacity has been raised, repeat forward (+ backward)."""
        if not self._pending:
            return True
        self._pending = False
        n = C.c_int(0)
        rc = _lib.lib().frg_forward_finish(_p(self.img.buf), 0, C.byref(n))
        if rc == _lib.ECAPACITY:
            self.capacity = int(n.value * self.capacity_slack) + 4096
            return False
        if rc < 0:
            raise RuntimeError(f"frg_forward_finish failed ({rc}): {_lib.last_error()}")
        self.true_num_rendered = n.value
        return True

    def backward(self, dL_dimage, slot: int = 0, payload=None, phase: int = 0, slot_sums: bool = False, local: bool = False,
                 sum_range=None):
        """Gradients of the last forward, written in place into exchange buffer `slot`.  In the
        factored plan under a process group, views["shs"] is only valid after wait_exchange(slot).
        payload: also fill this view's share of the factored exchange (masked colour gradient and
        camera centre); default: only when a process group is live.
        live_rows = True: ONLY the rows of the Gaussians marked in self.row_live are written -- every other row of the returned
        views, of dL_dmeans2D (the viewspace gradient) and of dL_dcolors holds whatever an earlier step left there.  The one
        consumer that may take the buffer as it is is FlatAdam.step(flat, row_live=self.row_live) (it rejects the buffer
        without the mask); anything else -- densification statistics, gradient norms, a plain optimizer -- calls
        zero_dead_rows(slot) first.
        sum_range = (first, count), slot-sum plan only: phase 1 in pieces (frg_backward_args::range_first / range_count) -- the
        call with first == 0 runs the blend backward, every call reduces its range's slots.
        phase (frg_backward_args::phase): 1 = the backward blend + the per-Gaussian slot sums -- the payload of the
        factored exchange is complete when this call's kernels are, so its all-gather can be started before phase 2
        (backward_overlapped does that); 2 = the rest; 0 = both in one call."""
        L = _lib.lib()
        s = self.scene
        cam, bg = self._view
        H, W = cam.image_height, cam.image_width
        self.exchange = ex = self.exchanges[slot]
        g = ex.views
        ex.flat._frg_rows_partial = bool(self.live_rows)     # (FlatAdam.step refuses such a buffer without its row mask)
        # (slot_sums=True: one process playing every rank -- tests, tools; local=True: no exchange follows this backward --
        # the whole backward of the own view, as without a process group)
        slot_sums = ex.slotsum and not local and (ex._active() or slot_sums)
        if sum_range is not None and not slot_sums:
            raise RuntimeError("sum_range: pieces of phase 1 exist on the slot-sum path only")
        if sum_range is None or sum_range[0] == 0:
            self._bwd_gen = getattr(self, "_bwd_gen", 0) + 1        # (the workspace now holds THIS backward's sums)
        if slot_sums:
            # slot-sum plan with live collectives: only phase 1 runs here -- the blend backward and the per-Gaussian slot sums;
            # the sums travel, and the per-Gaussian chain runs for every view's rows in the exchange's combine pass
            if phase == 2:
                return g
            phase = 1
        # factored plan with live collectives: the per-view SH rows are not materialised -- wait()
        # rebuilds their sum over views from the exchanged colour gradients
        defer_sh = (ex.factor_sh and ex._active()) or slot_sums
        ws = int(L.frg_backward_workspace_bytes(self.P, self.num_rendered))
        work = self.work.ensure(ws)
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        # every option of the backward in one struct (frg_backward_ex), the forward's arithmetic stated
        v = lambda t: None if t is None else t.data_ptr()
        raw = self.raw_params
        a = _lib.BackwardArgs(
            struct_size=C.sizeof(_lib.BackwardArgs), P=self.P, D=s.sh_degree, M=self.K, R=self.num_rendered, background=v(bg),
            width=W, height=H, means3D=v(s.means3D), shs=v(s.shs), colors_precomp=None, scales=None if raw else v(s.scales), scale_modifier=1.0,
            rotations=None if raw else v(s.rotations), cov3D_precomp=None, viewmatrix=v(cam.viewmatrix), projmatrix=v(cam.projmatrix),
            campos=v(cam.campos), tan_fovx=float(cam.tanfovx), tan_fovy=float(cam.tanfovy), radii=v(self.radii),
            geom_buffer=v(self.geom.buf), binning_buffer=v(self.binning.buf), image_buffer=v(self.img.buf),
            dL_dpix=v(dL_dimage), dL_dmean2D=v(self.dL_dmeans2D), dL_dconic=None, dL_dopacity=v(g["opacities"]),
            dL_dcolor=v(ex.own_drgb) if defer_sh else (v(self.dL_dcolors) if (ex.factor_sh or self.write_all_outputs) else None),
            dL_dmean3D=v(g["means3D"]), dL_dcov3D=v(self.dL_dcov3D), dL_dsh=None if defer_sh else v(g["shs"]),
            dL_dscale=v(g["scales"]), dL_drot=v(g["rotations"]), workspace=v(work), workspace_bytes=work.numel(), debug=0,
            hip_stream=stream.value, raw_opacities=v(s.opacities) if raw else None, raw_scales=v(s.scales) if raw else None,
            raw_rotations=v(s.rotations) if raw else None, exact_blend=getattr(self, "_exact", -1) + 1, phase=int(phase),
            row_live=v(self.row_live), range_first=0 if sum_range is None else int(sum_range[0]),
            range_count=0 if sum_range is None else int(sum_range[1]))
        rc = L.frg_backward_ex(C.byref(a))
        if rc < 0:
            raise RuntimeError(f"frg_backward failed ({rc}): {_lib.last_error()}")
        if slot_sums:
            gen = self._bwd_gen
            ex.note_view(work=work, R=self.num_rendered, cam=cam, D=s.sh_degree, scale_modifier=1.0,
                         intact=lambda: self._bwd_gen == gen)
            return g
        if phase == 2:
            return g              # (the payload left with phase 1)
        if ex.factor_sh and (ex._active() if payload is None else payload):
            # this view's share of the factored SH exchange: masked colour gradient + camera centre
            if not defer_sh:   # (with deferred SH rows the backward wrote the payload itself)
                rc = L.frg_sh_color_grad(self.P, _p(self.geom.buf), _p(self.radii), _p(self.dL_dcolors), _p(ex.own_drgb), stream)
                if rc < 0:
                    raise RuntimeError(f"frg_sh_color_grad failed ({rc}): {_lib.last_error()}")
            # 12 bytes, copied every time: a (pointer, version) tag cannot tell a fresh camera tensor that
            # the caching allocator placed at the previous one's address from the previous one
            ex.own_campos.copy_(cam.campos.reshape(-1)[:3], non_blocking=True)
        return g

    def zero_dead_rows(self, slot: int = 0):
        """live_rows mode: make the gradients of buffer `slot` (and the rank-local dL_dmeans2D / dL_dcolors / dL_dcov3D) DENSE
        -- the rows of Gaussians the last backward did not mark are filled with zeros, which is what the dense form writes
        there -- for consumers other than the masked optimizer step.  Costs what the mode saved; returns the views."""
        ex = self.exchanges[slot]
        if not self.live_rows:
            return ex.views
        dead = self.row_live == 0
        for t in list(ex.views.values()) + [self.dL_dmeans2D, self.dL_dcolors] + ([self.dL_dcov3D] if self.dL_dcov3D is not None else []):
            t[dead] = 0.0
        ex.flat._frg_rows_partial = False
        return ex.views

    def backward_overlapped(self, dL_dimage, slot: int = 0):
        """backward + the start of the exchange, factored plan: phase 1 (blend backward + slot sums: the colour-gradient
        payload is complete) -> the all-gather of the payloads is enqueued -> phase 2 (the per-Gaussian chain, 0.3 ms at
        C3) runs while they travel -> the sum of the dense part is enqueued.  Finish with
        exchanges[slot].finish_in_step() (or exchange_in_step(slot, started=True)).  Without a live factored exchange:
        a plain backward + start_exchange."""
        ex = self.exchanges[slot]
        if ex.slotsum and ex._active() and len(ex.chunks) > 1 and self.phase1_in_pieces:
            # slot-sum plan: phase 1 in pieces -- the blend backward, then per chunk of Gaussians: its slot sums, its packet, its
            # all-gather: chunk k travels while chunk k + 1 is still being reduced (and, in finish_in_step, while chunk k - 1
            # is being combined)
            self.exchange = ex
            ex._works = []
            for c, (first, n) in enumerate(ex.chunks):
                g = self.backward(dL_dimage, slot, sum_range=(first, n))
                ex.start_chunk(c)
            return g
        if not (ex.factor_sh and ex._active()) or ex.sparse:      # (sparse: one call -- its rows are packed from the finished gradients; slotsum: backward() runs phase 1 only)
            g = self.backward(dL_dimage, slot)
            ex.start()
            return g
        self.backward(dL_dimage, slot, phase=1)
        ex.start(part="gather")
        g = self.backward(dL_dimage, slot, phase=2)
        ex.start(part="dense")
        return g

    def exchange_in_step(self, slot: int = 0, started: bool = False):
        """start_exchange + GradientExchange.finish_in_step: the gradients of buffer `slot` are the sums over ranks
        when this returns (stream-ordered) -- the form a step with an optimizer update between views needs."""
        ex = self.exchanges[slot]
        if not started:
            ex.start()
        return ex.finish_in_step()

    def allreduce_grads(self, slot: int = 0):
        """Synchronous form: SUM over ranks, stream-ordered."""
        return self.exchanges[slot].all_reduce()

    def start_exchange(self, slot: int):
        """Launch the all-reduce of buffer `slot` behind its backward; returns immediately."""
        return self.exchanges[slot].start()

    def prefetch_exchange(self, slot: int):
        """Finish the pending exchange of buffer `slot` with the SH rebuild on a side stream (see
        GradientExchange.wait_on_side_stream); the dense gradients may be overwritten right away, a
        later wait_exchange(slot) joins the rebuild.  Call between forward() and backward(.., slot)."""
        self.exchanges[slot].wait_on_side_stream()

    def wait_exchange(self, slot: int):
        """Order the current stream after the pending all-reduce of buffer `slot` (call before
        reading its gradients or before the next backward that overwrites it)."""
        return self.exchanges[slot].wait()

"""CPU tests of the boundary: C-ABI library loads and exports every declared symbol,
the Python host mirror keeps the reference's names / argument rules / errors."""
import os
import re

import pytest
import torch

from frosting_amd import _lib
from frosting_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _settings():
    e = torch.eye(4)
    return GaussianRasterizationSettings(64, 64, 0.5, 0.5, torch.zeros(3), 1.0, e, e, 3, torch.zeros(3), False, False)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "frosting_rasterizer.h")).read()
    declared = set(re.findall(r"\b(frg_[a-z_0-9]+)\s*\(", hdr)) - {"frg_alloc_fn"}
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.frg_version() == 2


def test_state_size_queries_and_options():
    L = _lib.lib()
    assert L.frg_geometry_bytes(1000) >= 1000 * 56
    assert L.frg_geometry_bytes(2000) > L.frg_geometry_bytes(1000)
    assert L.frg_image_bytes(1600, 1056) >= 1600 * 1056 * 8
    assert L.frg_binning_bytes(1000, 10) >= 1000 * 12
    assert L.frg_binning_bytes(1000, 100000) > L.frg_binning_bytes(1000, 10)  # ping-pong buffer for oversize tiles
    assert L.frg_backward_workspace_bytes(10, 1000) >= 1000 * 36
    old = _lib.set_option("exact_blend", 1)
    assert _lib.get_option("exact_blend") == 1
    _lib.set_option("exact_blend", old)
    # every documented option answers get / set and returns the previous value (include/frosting_rasterizer.h)
    for name, default in (("tight_binning", 0), ("global_bins", 0), ("bwd_waves", 0), ("bwd_seg_log", 0), ("fwd_order", 1), ("counter_mailbox", 1),
                          ("clear_image_state", 0), ("sort_heavy_on_caller", 1), ("bwd_heavy_first", 1), ("fwd_prefetch", 1), ("sparse_sh", 1), ("sh_dir_in_backward", 0), ("profile", 0), ("profile_stage", -1)):
        assert _lib.get_option(name) == default, name
        assert _lib.set_option(name, default) == default and _lib.get_option(name) == default, name
    assert _lib.set_option("no_such_option", 1) < 0 and "unknown option" in _lib.last_error()
    # the timing-experiment knobs (kernels skip work: wrong results) only exist under FROSTING_EXPERIMENTS=1
    if os.environ.get("FROSTING_EXPERIMENTS") != "1":
        for knob in ("ablate", "probe", "rows_grid"):
            assert _lib.set_option(knob, 1) < 0 and "FROSTING_EXPERIMENTS" in _lib.last_error()


def test_settings_fields_match_reference_order():
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")


def test_argument_validation_like_reference():
    r = GaussianRasterizer(raster_settings=_settings())
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3), colors_precomp=torch.zeros(4, 3),
          scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3), scales=torch.ones(4, 3))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3), scales=torch.ones(4, 3),
          rotations=torch.ones(4, 4), cov3D_precomp=torch.zeros(4, 6))


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU tensor -- never route through a CPU path."""
    r = GaussianRasterizer(raster_settings=_settings())
    m = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3), scales=torch.ones(4, 3),
          rotations=torch.ones(4, 4))
    with pytest.raises(RuntimeError, match="num_points, 3"):
        from frosting_amd.rasterizer import _C
        _C.rasterize_gaussians(torch.zeros(3), torch.zeros(4, 2), *([torch.Tensor([])] * 4), 1.0, torch.Tensor([]),
                               torch.eye(4), torch.eye(4), 0.5, 0.5, 8, 8, torch.Tensor([]), 0, torch.zeros(3),
                               False, False)


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "frosting_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "gs_oracle" not in src and "ref_rasterizer" not in src and "from oracle" not in src, f
    drop_in = open(os.path.join(ROOT, "diff_gaussian_rasterization", "__init__.py")).read()
    assert "oracle" not in drop_in


def test_drop_in_import_name():
    """The reference's package name serves the same host API over the COMPILED torch extension
    (diff_gaussian_rasterization._C, built by setup.py), not over the ctypes binding."""
    import diff_gaussian_rasterization as d
    assert issubclass(d.GaussianRasterizer, GaussianRasterizer) and d.GaussianRasterizer.__name__ == "GaussianRasterizer"
    assert d.GaussianRasterizationSettings is GaussianRasterizationSettings
    assert {"rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"} <= set(dir(d._C))   # DGR/ext.cpp:15-18
    assert type(d._C).__name__ == "module" and d._C.__file__.endswith(".so") and d.GaussianRasterizer._ops is d._C
    assert d._C.library_version() == _lib.lib().frg_version()
    # same argument validation on the compiled path, before any device work (rasterize_points.cu:57-59)
    e = torch.Tensor([])
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        d._C.rasterize_gaussians(e, torch.zeros(4, 2), e, e, e, e, 1.0, e, e, e, 1.0, 1.0, 8, 8, e, 0, e, False, False)
    with pytest.raises(RuntimeError, match="no CPU path"):
        d._C.rasterize_gaussians(e, torch.zeros(4, 3), e, e, e, e, 1.0, e, e, e, 1.0, 1.0, 8, 8, e, 0, e, False, False)


def test_extended_entry_points_validate_arguments_without_a_gpu():
    """Argument checks of the additive entry points run before any HIP call."""
    import ctypes as C
    L = _lib.lib()
    assert C.sizeof(_lib.ForwardArgs) % 8 == 0
    a = _lib.ForwardArgs()
    a.struct_size = 8                                     # a caller built against another header
    assert L.frg_forward_ex(C.byref(a)) == -1 and "struct_size" in _lib.last_error()
    assert L.frg_forward_ex(None) == -1
    # four generations of frg_forward_args pass the size check (then fail on their null pointers, not on the size);
    # forward_only is 0 | 1 and is not offered with deferred counters
    for field in ("raw_opacities", "exact_blend", "forward_only"):
        a = _lib.ForwardArgs()
        a.struct_size = getattr(_lib.ForwardArgs, field).offset
        assert L.frg_forward_ex(C.byref(a)) < 0 and "struct_size" not in _lib.last_error(), field
    a = _lib.ForwardArgs()
    a.struct_size = C.sizeof(_lib.ForwardArgs)
    a.forward_only = 2
    assert L.frg_forward_ex(C.byref(a)) == -1 and "forward_only" in _lib.last_error()
    a.forward_only, a.instance_capacity = 1, 1000
    assert L.frg_forward_ex(C.byref(a)) == -1 and "deferred" in _lib.last_error()
    assert _lib.mode_fields({"forward_only": 1, "exact_blend": 1}) == {"exact_blend": 2, "tight_binning": 0, "async_sh": 0, "forward_only": 1}
    # frg_backward_args: five generations, told apart by struct_size (up to shell_*, + exact_blend / shell_bary_mode,
    # + phase, + row_live, + range_first / range_count); the ctypes mirror is the newest.  P == 0 returns before any pointer is looked at.
    b = _lib.BackwardArgs(P=0, width=8, height=8)
    for size in (C.sizeof(_lib.BackwardArgs), _lib.BackwardArgs.range_first.offset, _lib.BackwardArgs.row_live.offset,
                 _lib.BackwardArgs.phase.offset, _lib.BackwardArgs.exact_blend.offset):
        b.struct_size = size
        assert L.frg_backward_ex(C.byref(b)) == 0, (size, _lib.last_error())
    b.struct_size = C.sizeof(_lib.BackwardArgs) + 8
    assert L.frg_backward_ex(C.byref(b)) == -1 and "struct_size" in _lib.last_error()
    b.struct_size = C.sizeof(_lib.BackwardArgs)
    b.P, b.phase = 5, 3
    assert L.frg_backward_ex(C.byref(b)) == -1      # (null pointers or the phase: refused either way, before any HIP call)
    # the workspace covers the slots and the per-Gaussian sums of a two-call backward; the backward blend's work items are
    # listed by the forward in its own chunks (8 bands x R / 256 full-segment items behind point_list and pairs, the
    # checkpoints behind them: 16 / 8 B per instance for segments of 256 / 512 entries)
    assert L.frg_backward_workspace_bytes(1000, 100_000) >= 100_000 * 36 + 1000 * 36
    assert L.frg_binning_bytes(100_000, 10) >= 100_000 * (4 + 8 + 16) + 8 * (100_000 // 256) * 8
    assert L.frg_binning_bytes(1 << 24, 10) >= (1 << 24) * (4 + 8 + 8) + 8 * ((1 << 24) // 256) * 8
    assert L.frg_binning_bytes(1 << 24, 10) < (1 << 24) * (4 + 8 + 9 + 1)
    n = C.c_int(-7)
    assert L.frg_forward_finish(None, 0, C.byref(n)) == -1 and "pending" in _lib.last_error()
    # deferred forward needs a positive capacity
    args = [_lib.ALLOC_FN(lambda u, b: 0)] * 3 + [None, 4, 0, 0, None, 8, 8] + [None] * 4 + [None, 1.0, None, None] + \
           [None] * 3 + [0.5, 0.5, 0, None, None, 0, None]
    assert L.frg_forward_deferred(*args) == -1 and "instance_capacity" in _lib.last_error()
    # SH rebuild: degree / coefficient-count checks
    assert L.frg_sh_grad_from_views(4, 4, 16, 1, None, None, 0, None, 0, None, None) == -1
    assert L.frg_sh_grad_from_views(4, 3, 9, 1, None, None, 0, None, 0, None, None) == -1 and "coefficients" in _lib.last_error()
    assert L.frg_sh_grad_from_views(0, 3, 16, 0, None, None, 0, None, 0, None, None) == 0      # nothing to do
    assert L.frg_sh_color_grad(0, None, None, None, None, None) == 0
    assert L.frg_sh_color_grad(4, None, None, None, None, None) == -1


def test_adam_entry_point_validates_arguments_without_a_gpu():
    import ctypes as C
    L = _lib.lib()
    ends, lrs = (C.c_longlong * 2)(4, 8), (C.c_float * 2)(0.1, 0.2)
    assert L.frg_adam_step(8, None, None, None, None, ends, lrs, None, None, None, 2, 0.9, 0.999, 1e-15, 1, 1.0, None) == -1   # null arrays
    assert L.frg_adam_step(9, None, None, None, None, ends, lrs, None, None, None, 2, 0.9, 0.999, 1e-15, 1, 1.0, None) == -1 \
        and "end at n" in _lib.last_error()
    assert L.frg_adam_step(8, None, None, None, None, ends, lrs, None, None, None, 9, 0.9, 0.999, 1e-15, 1, 1.0, None) == -1
    assert L.frg_adam_step(8, None, None, None, None, ends, lrs, None, None, None, 2, 0.9, 0.999, 1e-15, 0, 1.0, None) == -1   # step is 1-based
    ends0 = (C.c_longlong * 1)(0)
    assert L.frg_adam_step(0, None, None, None, None, ends0, lrs, None, None, None, 1, 0.9, 0.999, 1e-15, 1, 1.0, None) == 0    # nothing to do


def test_bench_self_launches_n_ranks_from_a_bare_shell(monkeypatch):
    """`python bench.py --gpus 8` without torch.distributed.run around it must become 8 ranks
    (the driver's N=1 form, applied to N>1, used to run ONE rank)."""
    import sys
    import bench
    seen = {}
    monkeypatch.setattr(bench.os, "execv", lambda exe, argv: seen.update(exe=exe, argv=argv))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7"])
    args = bench.parse_args()
    bench.self_launch_if_needed(args)
    a = seen["argv"]
    assert a[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in a and "--nnodes=1" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and a[-4:] == ["--gpus", "8", "--steps", "7"]
    # already under a launcher (WORLD_SIZE set), or N == 1: no re-exec
    seen.clear()
    monkeypatch.setenv("WORLD_SIZE", "8")
    bench.self_launch_if_needed(args)
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench.self_launch_if_needed(bench.parse_args())
    assert not seen


def test_ctypes_structs_have_the_headers_layout(tmp_path):
    """frosting_amd/_lib.py restates frg_forward_args / frg_backward_args field by field: every field of the ctypes
    structures sits at the offset the C compiler gives it in include/frosting_rasterizer.h, and the sizes agree (a field
    added to one side only would shift everything behind it silently)."""
    import ctypes as C
    import subprocess
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    lines = []
    structs = (("frg_forward_args", _lib.ForwardArgs), ("frg_backward_args", _lib.BackwardArgs), ("frg_combine_args", _lib.CombineArgs))
    for cname, ct in structs:
        lines.append(f'printf("{cname} sizeof %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "frosting_rasterizer.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", inc, str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    want = {}
    for ln in out:
        if ln.strip():
            cname, fname, val = ln.split()
            want[(cname, fname)] = int(val)
    for cname, ct in structs:
        assert C.sizeof(ct) == want[(cname, "sizeof")], cname
        for fname, _ in ct._fields_:
            assert getattr(ct, fname).offset == want[(cname, fname)], (cname, fname)


def test_sum_packet_size_formula():
    """frosting_amd.parallel.sum_packet_words (what the Python layer sizes its all-gather buffers with, on any backend) is the
    library's frg_sum_packet_bytes: header, one bit per Gaussian, one row offset per 64 Gaussians, 48-byte rows."""
    from frosting_amd.parallel import sum_packet_words
    L = _lib.lib()
    for n, cap in ((1, 0), (64, 64), (65, 7), (1000, 1000), (1_500_000, 204_800), (3_000_000, 3_000_000)):
        assert 4 * sum_packet_words(n, cap) == L.frg_sum_packet_bytes(n, cap), (n, cap)
    assert sum_packet_words(1_500_000, 204_800) * 4 < 0.15 * 1_500_000 * 48 + 400_000

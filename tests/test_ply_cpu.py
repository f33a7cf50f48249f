"""PLY checkpoints in the reference's layout (gaussian_model.py:178-256): header and attribute order,
channel-major SH features, raw parameters, round trip, and files written by other tools (ascii, other order)."""
import struct

import numpy as np
import pytest
import torch

from frosting_amd import ply


def _model(P=7, K=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(xyz=r(P, 3), features_dc=r(P, 1, 3), features_rest=r(P, K - 1, 3), opacity=r(P, 1), scaling=r(P, 3), rotation=r(P, 4))


def test_layout_matches_the_references_save_ply(tmp_path):
    m = _model()
    path = str(tmp_path / "sub" / "point_cloud.ply")
    ply.save_gaussians_ply(path, **m)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 7"]
    names = [ln.split()[2] for ln in lines[3:]]
    assert all(ln.startswith("property float ") for ln in lines[3:])
    assert names == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
        ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]          # :178-190
    assert len(body) == 7 * 62 * 4
    row0 = struct.unpack("<62f", body[: 62 * 4])
    assert row0[:3] == tuple(m["xyz"][0].tolist()) and row0[3:6] == (0.0, 0.0, 0.0)                # normals are zeros (:196)
    assert row0[6:9] == tuple(m["features_dc"][0, 0].tolist())
    # f_rest is channel-major: all 15 coefficients of R, then G, then B (transpose(1, 2).flatten, :198)
    assert row0[9:9 + 15] == pytest.approx(m["features_rest"][0, :, 0].tolist(), abs=0)
    assert row0[9 + 15:9 + 30] == pytest.approx(m["features_rest"][0, :, 1].tolist(), abs=0)
    assert row0[54] == m["opacity"][0, 0].item() and row0[58:62] == tuple(m["rotation"][0].tolist())


def test_round_trip_is_exact(tmp_path):
    for K in (16, 4, 1):
        m = _model(P=33, K=K, seed=K)
        path = str(tmp_path / f"k{K}.ply")
        ply.save_gaussians_ply(path, **m)
        deg = int(round(K ** 0.5)) - 1
        back = ply.load_gaussians_ply(path, max_sh_degree=deg)
        for k in m:
            assert back[k].shape == m[k].shape and torch.equal(back[k], m[k]), k
    with pytest.raises(ValueError, match="f_rest"):
        ply.load_gaussians_ply(path, max_sh_degree=3)                  # :230 asserts the count


def test_reads_ascii_and_reordered_properties(tmp_path):
    path = tmp_path / "a.ply"
    props = ["rot_3", "rot_2", "rot_1", "rot_0", "x", "y", "z", "opacity", "scale_0", "scale_1", "scale_2", "f_dc_0", "f_dc_1", "f_dc_2"]
    rows = np.arange(2 * len(props), dtype=np.float64).reshape(2, -1)
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\n" + "".join(f"property double {p}\n" for p in props))
        f.write("element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        for r in rows:
            f.write(" ".join(repr(float(x)) for x in r) + "\n")
    m = ply.load_gaussians_ply(str(path), max_sh_degree=0)
    assert m["rotation"].tolist() == [[3.0, 2.0, 1.0, 0.0], [17.0, 16.0, 15.0, 14.0]]
    assert m["xyz"][1].tolist() == [18.0, 19.0, 20.0] and m["features_rest"].shape == (2, 0, 3)
    assert m["features_dc"][0, 0].tolist() == [11.0, 12.0, 13.0]

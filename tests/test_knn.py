"""distCUDA2: the CPU restatement against scipy's k-d tree, the HIP search against the
restatement bit for bit."""
import numpy as np
import pytest
import torch

from oracle import knn_oracle as KO


def _cloud(n, seed, clustered=False):
    g = np.random.default_rng(seed)
    p = g.standard_normal((n, 3)).astype(np.float32)
    if clustered:
        p[: n // 2] = p[: n // 2] * 0.01 + 5.0                   # a dense clump far from the rest
        p[n // 2: n // 2 + 5] = p[0]                             # exact duplicates (distance 0)
    return p


def test_restatement_against_kdtree():
    from scipy.spatial import cKDTree
    p = _cloud(3000, 1, clustered=True)
    d, _ = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=4)
    want = (d[:, 1:] ** 2).mean(axis=1)
    got = KO.dist2_mean3(p)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-10)
    assert KO.dist2_mean3(p[:1])[0] == np.inf                     # no neighbours: FLT_MAX sums overflow, like the reference


@pytest.mark.gpu
@pytest.mark.parametrize("n,clustered", [(1, False), (2, False), (5, False), (255, False), (257, True), (5000, True), (20000, False)])
def test_hip_matches_restatement_bit_for_bit(gpu_device, n, clustered):
    from frosting_amd.knn import distCUDA2
    p = _cloud(n, n, clustered)
    got = distCUDA2(torch.from_numpy(p).to(gpu_device)).cpu().numpy()
    want = KO.dist2_mean3(p)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_large_cloud_properties_and_module_shim(gpu_device):
    """1 M points: a random sample against the brute-force restatement; the simple_knn import shim."""
    from frosting_amd.knn import install_as_simple_knn
    install_as_simple_knn()
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(5)
    pts = torch.randn(1_000_000, 3, generator=g)
    got = distCUDA2(pts.to(gpu_device)).cpu().numpy()
    assert got.shape == (1_000_000,) and np.isfinite(got).all() and (got > 0).all()
    p = pts.numpy()
    sel = np.random.default_rng(0).choice(len(p), 64, replace=False)
    for i in sel:
        d = ((p[i, 0] - p[:, 0]) ** 2 + (p[i, 1] - p[:, 1]) ** 2) + (p[i, 2] - p[:, 2]) ** 2
        d[i] = np.inf
        b = np.sort(np.partition(d, 2)[:3])
        assert got[i] == ((b[0] + b[1]) + b[2]) / np.float32(3.0)
    with pytest.raises(RuntimeError, match="GPU only"):
        distCUDA2(pts[:10])


def _ref_knn(points_dev):
    """The reference's own SimpleKNN::knn (oracle/_ref/libref_simple_knn.so, built from
    gaussian_splatting/submodules/simple-knn/simple_knn.cu by oracle/build_ref.sh)."""
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_simple_knn.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_simple_knn.so not built")
    L = C.CDLL(path)
    out = torch.zeros(points_dev.shape[0], dtype=torch.float32, device=points_dev.device)
    torch.cuda.synchronize()
    rc = L.ref_knn(C.c_int(points_dev.shape[0]), C.c_void_p(points_dev.data_ptr()), C.c_void_p(out.data_ptr()))
    assert rc == 0
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n,kind", [(4, "ball"), (1025, "clustered"), (100_000, "clustered"), (3_000_000, "c3"),
                                    (2_000_000, "shell")])
def test_hip_matches_the_reference_binary_bit_for_bit(gpu_device, n, kind):
    """PINNED: frg_knn_mean_dist2 against the reference's simple-knn compiled from its own source, on the
    3 M points of C3, the 2 M shell-bound points of C4 and clouds with duplicates / far clumps."""
    from frosting_amd import scenes
    from frosting_amd.knn import distCUDA2
    if kind == "c3":
        p = scenes.make_scene(n, scenes.CONFIGS["c3"]["seed"]).means3D
    elif kind == "shell":
        p = scenes.make_shell_scene(n, scenes.CONFIGS["c4"]["seed"]).scene.means3D
    else:
        p = torch.from_numpy(_cloud(n, n, kind == "clustered"))
    p = p.to(gpu_device).contiguous()
    got = distCUDA2(p)
    want = _ref_knn(p.clone())
    assert torch.equal(got, want), f"{int((got != want).sum())} of {n} differ"
    assert bool(torch.isfinite(got).all())

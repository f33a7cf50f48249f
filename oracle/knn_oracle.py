"""TEST INFRASTRUCTURE.  CPU restatement of simple-knn's distCUDA2
(gaussian_splatting/submodules/simple-knn/simple_knn.cu:147-198): for every point the three smallest squared
distances to the OTHER points, kept in ascending order by the reference's insertion (updateKBest, :147-160),
distance = dx*dx + dy*dy + dz*dz evaluated left to right in float32 (:150-151), result (b0 + b1 + b2) / 3.0f
(:197).  The Morton ordering and box pruning of the reference only decide WHICH pairs are skipped, never the
value, so a brute-force scan restates it exactly.

Parity unpinned against the reference binary (simple-knn needs CUB/Thrust and is not built here); the value
is fully specified by the lines cited, and tests cross-check this file against scipy's k-d tree in float64.
"""
import numpy as np


def dist2_mean3(points: np.ndarray, block: int = 512) -> np.ndarray:
    p = np.ascontiguousarray(points, dtype=np.float32)
    n = p.shape[0]
    out = np.empty(n, np.float32)
    fmax = np.float32(np.finfo(np.float32).max)
    for s in range(0, n, block):
        q = p[s:s + block]
        dx = q[:, None, 0] - p[None, :, 0]
        dy = q[:, None, 1] - p[None, :, 1]
        dz = q[:, None, 2] - p[None, :, 2]
        d = (dx * dx + dy * dy) + dz * dz                      # float32, left to right
        d[np.arange(q.shape[0]), np.arange(s, s + q.shape[0])] = np.inf   # i == idx is skipped (:177,191)
        k = min(3, n - 1)
        best = np.full((q.shape[0], 3), fmax, np.float32)
        if k > 0:
            part = np.partition(d, k - 1, axis=1)[:, :k]
            best[:, :k] = np.sort(part, axis=1)
        with np.errstate(over="ignore"):
            out[s:s + block] = ((best[:, 0] + best[:, 1]) + best[:, 2]) / np.float32(3.0)
    return out

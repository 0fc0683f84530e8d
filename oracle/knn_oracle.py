"""CPU oracle of distCUDA2 -- TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu legs).

Restates the RESULT the reference's simple-knn defines (submodules/simple-knn/simple_knn.cu:129-183: updateKBest<3> over
every j != i, then (best[0]+best[1]+best[2])/3 at :182): the three smallest float32 values of
(xj-xi)^2 + (yj-yi)^2 + (zj-zi)^2 over all other points, FLT_MAX for unfilled slots.  The Morton order / boxes of the
reference (:185-221) only prune and are not restated.  PARITY UNPINNED against an execution of the reference: simple-knn is
CUDA-only and cannot be built here, and the reference holds no test vectors for it; the oracle is cross-checked by two
independent formulations instead (dense brute force, and scipy cKDTree candidate search + float32 re-evaluation).
"""
import numpy as np

f32 = np.float32
FLT_MAX = np.finfo(np.float32).max


def _d2(q, c):
    d = c - q                                            # point - ref, simple_knn.cu:131
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def _mean3(best):
    with np.errstate(over="ignore"):
        return ((best[:, 0] + best[:, 1]) + best[:, 2]) / f32(3.0)


def dist2_bruteforce(points, chunk=512):
    p = np.ascontiguousarray(points, dtype=f32)
    P = p.shape[0]
    out = np.empty(P, f32)
    for s in range(0, P, chunk):
        q = p[s:s + chunk]
        d = _d2(q[:, None, :], p[None, :, :]).astype(f32)
        d[np.arange(q.shape[0]), np.arange(s, s + q.shape[0])] = FLT_MAX            # j != i by index
        k = min(3, P)
        best = np.full((q.shape[0], 3), FLT_MAX, f32)
        part = np.partition(d, k - 1, axis=1)[:, :k] if P > k else d
        best[:, :part.shape[1]] = np.sort(part, axis=1)[:, :3]
        out[s:s + chunk] = _mean3(best)
    return out


def dist2_kdtree(points, k_search=12):
    """Candidate search in double precision (cKDTree), distances re-evaluated with the float32 expression; the extra
    candidates absorb float32 re-ranking of near-ties."""
    from scipy.spatial import cKDTree
    p = np.ascontiguousarray(points, dtype=f32)
    P = p.shape[0]
    k = min(k_search + 1, P)
    _, idx = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=k)
    idx = idx.reshape(P, k)
    d = _d2(p[:, None, :], p[idx]).astype(f32)
    d[idx == np.arange(P)[:, None]] = FLT_MAX
    best = np.full((P, 3), FLT_MAX, f32)
    srt = np.sort(d, axis=1)[:, :3]
    best[:, :srt.shape[1]] = srt
    return _mean3(best)

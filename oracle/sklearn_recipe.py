"""numpy restatement of the reference's anchor clustering call

    sklearn.cluster.KMeans(n_clusters=S, random_state=0, init='k-means++', n_init=10).fit(C.T)
                                                            (EigenTrajectory/anchor.py:65-71)

TEST INFRASTRUCTURE ONLY (see oracle/et_oracle.c).  scikit-learn is a third-party dependency of the
reference that is not under /root/reference (version unpinned by the repo; the golden fixtures were
captured with 1.7.2, tests/golden/MANIFEST.json).  This file restates its published algorithm in the
arithmetic of its float32 code path and is pinned against sklearn's OWN outputs captured in the build
container (tests/golden/g11_sklearn_anchors.npz, made by tools/make_golden_sklearn.py):

* sklearn/cluster/_kmeans.py  KMeans.fit:  X -= X.mean(axis=0);  tol = mean(var(X, axis=0)) * 1e-4
  (numpy adds the rows of a C-ordered float32 (N,d) array sequentially in float32);
* _kmeans_plusplus: first centre = RandomState.choice(n) (uniform p), then per centre 2+log(K) candidates
  by D^2 sampling -- searchsorted(stable_cumsum(closest), uniform * potential) -- and the candidate with the
  smallest resulting potential wins;
* squared distances of float32 data: float64 ((-2 x.c) + |c|^2) + |x|^2, cast to float32, clamped at 0
  (sklearn/metrics/pairwise.py _euclidean_distances_upcast);
* Lloyd iterations until sum((c - c')^2) <= tol or max_iter = 300, best of n_init by final inertia.

Not restated bit-for-bit (and why the pin is on seed indices + centre / ADE / FDE closeness, not on bits):
the potential is a float32 BLAS dot in sklearn (summation order unspecified; here the float64 sum rounded to
float32), Lloyd sums are float32 per thread chunk in sklearn (here the oracle's exact sums).  Empty clusters: sklearn re-seeds each one, in
every iteration, with the point that is farthest from its own centre (_k_means_common.pyx
_relocate_empty_clusters_dense) -- restated in lloyd_relocating(), the loop an initialisation takes when the plain fit
(kmeans.py:182 semantics: 0/0 = NaN) ran into one; pinned by g11 "dup15" (more clusters than distinct points).
"""
from __future__ import annotations

import numpy as np

from . import et_oracle as eo


def center_columns(C, rel_tol=1e-4):
    """C (d,N) float32 -> (Xc (d,N) float32, mean (d,), tol float32) like KMeans.fit's pre-processing."""
    X = np.ascontiguousarray(np.asarray(C, np.float32).T)  # (N,d) C-ordered like sklearn's validated copy
    tol = np.float32(np.mean(np.var(X, axis=0)) * rel_tol)  # _check_params_vs_input: from the data AS GIVEN ...
    mean = X.mean(axis=0)          # sequential float32 adds over the rows
    X = X - mean                   # ... then centred (KMeans.fit)
    return np.ascontiguousarray(X.T), mean, tol


def seeding_uniforms(rng, K, n_init=1):
    """The draws sklearn's seeding consumes from a RandomState, one row per initialisation:
    [first-centre draw, (K-1) x n_trials thresholds]."""
    n_trials = 2 + int(np.log(K))
    U = np.empty((n_init, 1 + (K - 1) * n_trials), np.float64)
    for i in range(n_init):
        U[i, 0] = rng.random_sample()
        for c in range(1, K):
            U[i, 1 + (c - 1) * n_trials:1 + c * n_trials] = rng.uniform(size=n_trials)
    return U


def _dist(X64, xx, idx):
    Y = X64[:, idx]                                              # (d, m)
    d = (-2.0 * (Y.T @ X64) + (Y * Y).sum(axis=0)[:, None]) + xx[None, :]
    return np.maximum(d.astype(np.float32), np.float32(0))


def kmeanspp_seed(X, K, uniforms):
    """Greedy k-means++ on X (d,N) float32 with the given draws -> (centers (d,K) float32, indices (K,) int64)."""
    X = np.asarray(X, np.float32)
    d, n = X.shape
    n_trials = 2 + int(np.log(K))
    X64 = X.astype(np.float64)
    xx = (X64 * X64).sum(axis=0)
    first = min(int(uniforms[0] * n), n - 1)
    idx = [first]
    closest = _dist(X64, xx, [first])[0]
    pot = np.float32(closest.astype(np.float64).sum())
    for c in range(1, K):
        t = uniforms[1 + (c - 1) * n_trials:1 + c * n_trials] * np.float64(pot)
        cand = np.searchsorted(np.cumsum(closest, dtype=np.float64), t)
        np.clip(cand, None, n - 1, out=cand)
        D = np.minimum(closest[None, :], _dist(X64, xx, cand))
        pots = D.astype(np.float64).sum(axis=1).astype(np.float32)
        b = int(np.argmin(pots))
        pot, closest = pots[b], D[b]
        idx.append(int(cand[b]))
    idx = np.asarray(idx, np.int64)
    return np.ascontiguousarray(X[:, idx]), idx


def relocate_empty(Xrows, old_centers, labels, sums, counts, frac):
    """_relocate_empty_clusters_dense on exact integer sums: Xrows (N,d) float32, old_centers (d,K), labels (N,), sums (d,K)
    int64 with `frac` fractional bits, counts (K,) int64 -- both modified in place.  The points picked are numpy's own
    ``argpartition(distances, -n_empty)[:-n_empty-1:-1]`` on the float32 distances, like sklearn."""
    empty = np.where(counts == 0)[0]
    if empty.size == 0:
        return False
    dist = ((Xrows - np.ascontiguousarray(old_centers.T)[labels]) ** 2).sum(axis=1)
    far = np.argpartition(dist, -empty.size)[:-empty.size - 1:-1]
    for new, f in zip(empty, far):
        old = labels[f]
        fx = np.trunc(np.ldexp(Xrows[f].astype(np.float64), int(frac))).astype(np.int64)
        sums[:, old] -= fx
        sums[:, new] = fx
        counts[new] = 1
        counts[old] -= 1
    return True


def lloyd_relocating(X, c0, max_iter, tol):
    """sklearn's Lloyd loop INCLUDING the relocation of empty clusters, on the oracle's exact-sum steps.
    -> (centers (d,K) float32, iterations)."""
    X = np.asarray(X, np.float32)
    d, n = X.shape
    K = c0.shape[1]
    Xrows = np.ascontiguousarray(X.T)
    cen = np.ascontiguousarray(c0, np.float32)
    mx = float(np.abs(X).max())
    frac = eo.kmeans_frac_bits(mx, n)
    it = 0
    for it in range(1, max_iter + 1):
        sf = eo.kmeans_sim_frac_bits(mx, float(np.abs(cen[np.isfinite(cen)]).max()), d, n)
        labels, sums, counts, ss, nn = eo.kmeans_assign_accumulate(X, cen, frac, sf)
        relocate_empty(Xrows, cen, labels, sums, counts, frac)
        cen, err, _, done = eo.kmeans_update(sums, counts, ss, nn, n, frac, sf, float(tol), cen)
        if done:
            break
    return cen, it


def kmeans(C, K, random_state=0, n_init=10, max_iter=300, rel_tol=1e-4):
    """The whole call -> dict(centers (d,K) float32 incl. the mean, inertia (sum of squared distances of the
    centred data to the final centres), seeds (n_init,K), inertias (n_init,), best)."""
    X, mean, tol = center_columns(C, rel_tol)
    U = seeding_uniforms(np.random.RandomState(random_state), K, n_init)
    best, seeds, inertias = None, [], []
    for i in range(n_init):
        c0, idx = kmeanspp_seed(X, K, U[i])
        seeds.append(idx)
        res = eo.kmeans_fit(X, c0, max_iter, float(tol))
        cen = res["centroids"]
        if not np.isfinite(cen).all():  # an empty cluster on the way: sklearn's loop re-seeds it
            cen, _ = lloyd_relocating(X, c0, max_iter, tol)
        inertia = float(-eo.kmeans_assign(X, cen)[1].astype(np.float64).sum()) if np.isfinite(cen).all() else np.nan
        inertias.append(inertia)
        if np.isfinite(inertia) and (best is None or inertia < inertias[best]):
            best, best_cen = i, cen
    if best is None:
        raise RuntimeError("every initialisation produced an empty cluster")
    return dict(centers=(best_cen + mean[:, None]).astype(np.float32), inertia=inertias[best], seeds=np.stack(seeds),
                inertias=np.asarray(inertias), best=best, tol=tol, mean=mean)

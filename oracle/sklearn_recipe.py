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
float32), Lloyd sums are float32 per thread chunk in sklearn (here the oracle's exact sums), empty clusters are
re-seeded by sklearn (here: that initialisation is discarded).
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
        inertia = float(-eo.kmeans_assign(X, cen)[1].astype(np.float64).sum()) if np.isfinite(cen).all() else np.nan
        inertias.append(inertia)
        if np.isfinite(inertia) and (best is None or inertia < inertias[best]):
            best, best_cen = i, cen
    if best is None:
        raise RuntimeError("every initialisation produced an empty cluster")
    return dict(centers=(best_cen + mean[:, None]).astype(np.float32), inertia=inertias[best], seeds=np.stack(seeds),
                inertias=np.asarray(inertias), best=best, tol=tol, mean=mean)

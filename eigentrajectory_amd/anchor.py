"""ETAnchor -- the reference's EigenTrajectory/anchor.py interface on HIP kernels.

Reference: EigenTrajectory/anchor.py:5-88.  Same parameter (``C_anchor`` (k,S)).
``anchor_generation`` clusters the projected coefficients on the GPU where the reference
calls ``sklearn.cluster.KMeans(n_clusters=S, random_state=0, init='k-means++', n_init=10)``
(anchor.py:65-71).  Two modes (``hyper_params.anchor_init``):

* ``"sklearn"`` (default -- the reference's semantics): the recipe of that sklearn call on the
  device, in the arithmetic of sklearn's float32 code path (csrc/et_kmeanspp.hip): mean-centred
  data, ``tol = 1e-4 * mean(var)``, ``max_iter = 300``, ten initialisations of greedy k-means++
  (D^2 sampling, 2 + log K candidates per centre) fed with the draws of ONE
  ``numpy.random.RandomState(0)`` exactly as sklearn consumes them, Lloyd iterations with exact sums
  (``et_kmeans_fit``), best final inertia wins.  On the ETH/UCY fit sets this draws the same seed
  points as sklearn and lands on the same anchors to ~1e-5 (tests/golden/g11, the own-fit ADE/FDE
  test).  sklearn itself is not bit-reproducible with more than one thread.
  Differences: an initialisation that produces an empty cluster is discarded (sklearn re-seeds the
  cluster; with k-means++ seeds it does not occur on the datasets).
* ``"farthest"``: this build's BatchKMeans (kmeans.py semantics) -- farthest-first seeding + Lloyd;
  one initialisation, deterministic, bit-identical to the CPU oracle, ~10x cheaper; the anchors are a
  different local optimum (ETH zero-predictor ADE/FDE 0.374/0.589 against the reference's 0.377/0.643).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .kmeans import BatchKMeans


class ETAnchor(nn.Module):
    r"""EigenTrajectory anchor model

    Args:
        hyper_params (DotDict): The hyper-parameters
    """

    def __init__(self, hyper_params):
        super().__init__()

        self.hyper_params = hyper_params
        self.k = hyper_params.k
        self.s = hyper_params.num_samples
        self.dim = hyper_params.traj_dim

        self.C_anchor = nn.Parameter(torch.zeros((self.k, self.s)))

    def to_ET_space(self, traj, evec):
        r"""Euclidean -> ET (anchor.py:22-36)"""
        tdim = evec.size(0)
        C, _, _, _ = ops.norm_project(traj.reshape(-1, tdim // 2, 2), None, None, None, evec, None,
                                      ops.MODE_IDENTITY, want_nrm=False, want_flag=False)
        return C

    def to_Euclidean_space(self, C, evec):
        r"""ET -> Euclidean (anchor.py:38-52)"""
        return ops.anchor_reconstruct(C.unsqueeze(-1), None, None, None, evec, ops.MODE_IDENTITY)[0]

    def anchor_generation(self, pred_traj_norm, U_pred_trunc, *, n_redo=1, max_iter=100, tol=1e-4, seed=0, mode=None):
        r"""Anchor generation on EigenTrajectory space (anchor.py:54-74; call once before training)."""
        C_pred = self.to_ET_space(pred_traj_norm, evec=U_pred_trunc)  # (k,N)
        self.generate_from_coefficients(C_pred, n_redo=n_redo, max_iter=max_iter, tol=tol, seed=seed, mode=mode)

    def generate_from_coefficients(self, C_pred, *, n_redo=1, max_iter=100, tol=1e-4, seed=0, mode=None):
        r"""Cluster ET coefficients (k,N) into S anchors.  ``mode``: "sklearn" | "farthest" (module docstring);
        default ``hyper_params.anchor_init`` if present, else "sklearn" (anchor.py:65-71).  ``n_redo``,
        ``max_iter`` and ``tol`` are BatchKMeans' arguments and apply to "farthest" only."""
        n = C_pred.shape[1]
        if n < self.s:
            raise ValueError(f"anchor generation needs at least num_samples={self.s} trajectories, got {n}")
        mode = mode or getattr(self.hyper_params, "anchor_init", None) or "sklearn"
        if mode == "sklearn":
            C_anchor, self.inertia_, self.seed_indices_ = sklearn_style_kmeans(C_pred, self.s, random_state=seed)
        elif mode == "farthest":
            km = BatchKMeans(n_clusters=self.s, n_redo=n_redo, max_iter=max_iter, tol=tol, init_mode="kmeans++")
            # kmeans.py:92 draws the first centroid from numpy's GLOBAL stream after the caller seeded it; here the same
            # draws come from a private stream (the moving / static fits of calculate_parameters run in two threads:
            # seeding and restoring the global state around a call that releases the GIL would race)
            km.rng = np.random.RandomState(seed)
            km.fit(C_pred[None].contiguous())
            C_anchor, self.inertia_ = km.centroids[0], km.inertia_
        else:
            raise ValueError(f"unknown anchor mode {mode!r}")
        # Register anchors as model parameters
        self.C_anchor = nn.Parameter(C_anchor.to(self.C_anchor.device))

    def forward(self, C_pred):
        r"""Anchor refinement on EigenTrajectory space (anchor.py:76-88); the wrapper fuses this into
        the reconstruction kernel and never materialises the sum."""
        return self.C_anchor.unsqueeze(dim=1).detach() + C_pred


# ------------------------------------------------------------------------------------------------
# The reference's sklearn call (anchor.py:65-71), restated on the device
# ------------------------------------------------------------------------------------------------
def seeding_uniforms(rng, K, n_init):
    """The draws sklearn's k-means++ consumes from ``rng`` (a ``numpy.random.RandomState``), one row per
    initialisation: one ``random_sample`` for the first centre (``choice`` with uniform p), then
    ``uniform(size=2 + log K)`` thresholds per further centre."""
    nt = ops.kmeanspp_trials(K)
    U = np.empty((n_init, 1 + (K - 1) * nt), dtype=np.float64)
    for i in range(n_init):
        U[i, 0] = rng.random_sample()
        for c in range(1, K):
            U[i, 1 + (c - 1) * nt:1 + c * nt] = rng.uniform(size=nt)
    return U


_UNIFORMS_CACHE = {}


def _lloyd_relocating(X, c0, max_iter, tol):
    """sklearn's Lloyd loop INCLUDING its treatment of empty clusters (sklearn/cluster/_k_means_common.pyx
    _relocate_empty_clusters_dense): in every iteration each empty cluster is re-seeded with the point that is farthest
    from its own centre, taken out of the cluster it was assigned to.  The fast fits (``et_kmeans_fit*``) keep
    BatchKMeans' semantics -- 0/0 = NaN, kmeans.py:182 -- so an initialisation that ran into an empty cluster is repeated
    here: the library's exact-sum steps on the device, the relocation (rare, a handful of points) on the host with the
    very numpy call sklearn makes (``argpartition`` of the float32 distances).  Does not happen with k-means++ seeds on
    ETH/UCY; it does when there are more clusters than distinct points.  -> final centres (d,K)."""
    dev = X.device
    d, n = X.shape
    K = c0.shape[1]
    sh = ops.KMeansShard(X, K)
    cen = c0.clone().contiguous()
    sh.scan()
    sh.begin(n, cen)
    frac = int(sh.read_state().frac)
    rows = None  # (the points come to the host when an empty cluster first appears: most repeated fits never need them)
    for _ in range(max_iter):
        part = sh.assign(cen)
        counts = part[d * K:d * K + K].cpu().numpy()  # K integers per iteration; everything else only on an empty cluster
        empty = np.where(counts == 0)[0]
        if empty.size:
            if rows is None:
                rows = np.ascontiguousarray(X.T.cpu().numpy())
            host = part.cpu().numpy().copy()
            counts = host[d * K:d * K + K]
            labels = sh.labels().cpu().numpy()
            old = np.ascontiguousarray(cen.T.cpu().numpy())
            dist = ((rows - old[labels]) ** 2).sum(axis=1)
            far = np.argpartition(dist, -empty.size)[:-empty.size - 1:-1]
            sums = host[:d * K].reshape(d, K)
            for new, f in zip(empty, far):
                src = labels[f]
                fx = np.trunc(np.ldexp(rows[f].astype(np.float64), frac)).astype(np.int64)
                sums[:, src] -= fx
                sums[:, new] = fx
                counts[new] = 1
                counts[src] -= 1
            part = torch.from_numpy(host).to(dev)
        sh.update(part, cen, tol)
        if sh.read_state().done:
            break
    return cen


def sklearn_style_kmeans(C, K, *, random_state=0, n_init=10, max_iter=300, tol=1e-4, concurrent=True):
    """``KMeans(n_clusters=K, random_state=random_state, init='k-means++', n_init=n_init).fit(C.T)`` as a recipe on
    the device: -> (cluster centres (d,K) fp32 on C's device, inertia = mean squared distance to the final centres,
    seed indices (n_init,K) int64 of every initialisation).

    The initialisations are independent problems on the same points.  ``concurrent`` (default): they are the batch
    dimension of the library's batched entry points -- the 4K-1 seeding launches serve all of them
    (``et_kmeanspp_seed_batch``), ONE persistent launch runs all their Lloyd loops (``et_kmeans_fit_batch``), one launch ranks them by the inertia of
    their final centres -- from the calling thread, on the caller's stream (ten host threads with a stream each, the
    round-2 form, scaled to barely 2x: stream / thread set-up and the fits slowing each other down).  ``concurrent=False``
    runs them one after the other.  The result does not depend on it: the best FINAL inertia wins, ties go to the
    earlier initialisation."""
    dev = ops.L.require_device(C)  # no CPU fallback: raises without a HIP device
    X, mean, tol_dev = ops.center_columns(C.to(device=dev, dtype=torch.float32), tol)
    d, n = X.shape
    key = (int(random_state), int(K), int(n_init))
    if key not in _UNIFORMS_CACHE:  # the draws of a seeding depend on (seed, K, n_init) only
        _UNIFORMS_CACHE[key] = seeding_uniforms(np.random.RandomState(random_state), K, n_init)
    U = torch.from_numpy(_UNIFORMS_CACHE[key]).to(dev)
    tol_ = float(tol_dev.item())

    if concurrent and n_init > 1:
        c0, seeds = ops.kmeanspp_seed_batch(X, K, U)
        res = ops.kmeans_fit_batch(X, c0, max_iter, tol_)
        cens = res["centroids"]                                      # (n_init, d, K)
        # an initialisation that ran into an empty cluster (NaN centre, kmeans.py:182 semantics): again, with sklearn's
        # re-seeding of empty clusters
        for b in torch.nonzero(~torch.isfinite(cens).flatten(1).all(dim=1)).flatten().tolist():
            cens[b] = _lloyd_relocating(X, c0[b], max_iter, tol_)
        _, maxsims = ops.kmeans_predict(X, cens)                       # every set of centres on the same points
        inertia = (-maxsims.double()).sum(dim=1)                     # sklearn ranks by the inertia of the FINAL centres
        best = int(torch.argmin(inertia))  # first minimum = the earlier initialisation on ties
        return (cens[best] + mean[:, None]).contiguous(), float(inertia[best]) / n, seeds

    nbytes = ops.L.lib().et_kmeanspp_workspace_bytes(ops.L.i64(n), d, ops.kmeanspp_trials(K))
    ws_seed = torch.empty((max(nbytes, 8),), device=dev, dtype=torch.uint8)
    best, seeds = None, []
    for i in range(n_init):
        c0, idx = ops.kmeanspp_seed(X, K, U[i], ws_seed)
        res = ops.kmeans_fit(X, c0, max_iter, tol_, trace=False)
        seeds.append(idx)
        cen = res["centroids"]
        if not bool(torch.isfinite(cen).all()):  # an empty cluster on the way: sklearn's loop re-seeds it
            cen = _lloyd_relocating(X, c0, max_iter, tol_)
        _, maxsims = ops.kmeans_predict(X, cen)  # sklearn ranks by the inertia of the FINAL centres
        inertia = float((-maxsims.double()).sum())
        if best is None or inertia < best[0]:
            best = (inertia, cen)
    return (best[1] + mean[:, None]).contiguous(), best[0] / n, torch.stack(seeds)

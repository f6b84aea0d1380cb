"""ETAnchor -- the reference's EigenTrajectory/anchor.py interface on HIP kernels.

Reference: EigenTrajectory/anchor.py:5-88.  Same parameter (``C_anchor`` (k,S)).
``anchor_generation`` clusters the projected coefficients on the GPU where the reference
calls ``sklearn.cluster.KMeans(n_clusters=S, random_state=0, init='k-means++', n_init=10)``
(anchor.py:65-71).  Two modes:

* ``"farthest"`` (default): this build's BatchKMeans -- farthest-first seeding + Lloyd with
  exact sums; deterministic for a given input and bit-identical to the CPU oracle.
* ``"sklearn"``: the recipe of the reference's sklearn call -- mean-centred data,
  ``tol = 1e-4 * mean(var)``, ``max_iter = 300``, ten initialisations of greedy k-means++
  (D^2 sampling, 2 + log K local trials) driven by ONE ``numpy.random.RandomState(0)`` stream
  exactly as sklearn consumes it, best inertia wins -- with the distance / potential passes and
  the Lloyd iterations on the device.  sklearn itself is not bit-reproducible with more than
  one thread, so this mode agrees with it in the seeds it draws and in quality
  (inertia), not in the last bits of the centres.  Differences: an empty cluster becomes NaN
  (kmeans.py:182) instead of being re-seeded (with k-means++ seeds this does not occur on the
  datasets), and the inertia that ranks the initialisations is the one of the last assignment.

Parity of the model is defined on loaded checkpoints (same anchors in -> same trajectories out).
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
        r"""Cluster ET coefficients (k,N) into S anchors (``mode``: "farthest" | "sklearn", see the module docstring;
        default: ``hyper_params.anchor_init`` if present, else "farthest")."""
        n = C_pred.shape[1]
        if n < self.s:
            raise ValueError(f"anchor generation needs at least num_samples={self.s} trajectories, got {n}")
        mode = mode or getattr(self.hyper_params, "anchor_init", None) or "farthest"
        if mode == "sklearn":
            C_anchor, self.inertia_, self.seed_indices_ = sklearn_style_kmeans(C_pred, self.s, random_state=seed)
            self.C_anchor = nn.Parameter(C_anchor.to(self.C_anchor.device))
            return
        if mode != "farthest":
            raise ValueError(f"unknown anchor mode {mode!r}")
        km = BatchKMeans(n_clusters=self.s, n_redo=n_redo, max_iter=max_iter, tol=tol, init_mode="kmeans++")
        state = np.random.get_state()
        try:
            np.random.seed(seed)  # the reference fixes random_state=0 (anchor.py:71); kmeans.py:92 draws from numpy
            km.fit(C_pred[None].contiguous())
        finally:
            np.random.set_state(state)
        C_anchor = km.centroids[0]
        self.inertia_ = km.inertia_
        # Register anchors as model parameters
        self.C_anchor = nn.Parameter(C_anchor.to(self.C_anchor.device))

    def forward(self, C_pred):
        r"""Anchor refinement on EigenTrajectory space (anchor.py:76-88); the wrapper fuses this into
        the reconstruction kernel and never materialises the sum."""
        return self.C_anchor.unsqueeze(dim=1).detach() + C_pred


# ------------------------------------------------------------------------------------------------
# The reference's sklearn call (anchor.py:65-71), restated on the device
# ------------------------------------------------------------------------------------------------
def greedy_kmeanspp(X, K, rng):
    """Greedy k-means++ seeding (Arthur & Vassilvitskii 2007 with 2 + log K local trials, the variant behind
    ``sklearn.cluster.KMeans(init='k-means++')``) on device points ``X`` (d,N) fp32.

    ``rng`` is a ``numpy.random.RandomState``; it is consumed exactly like sklearn consumes its stream (one
    uniform for the first centre, ``n_trials`` uniforms per further centre), so a shared stream stays aligned
    over several initialisations.  Distances / potentials / cumulative sums run on the device; only the
    ``n_trials`` random thresholds cross the PCIe bus per centre.  -> (indices list, centres (d,K)).
    """
    d, n = X.shape
    n_trials = 2 + int(np.log(K))
    first = min(int(rng.random_sample() * n), n - 1)  # choice(n, p=uniform): one uniform draw through the cdf
    idx = [first]
    closest = (-ops.euc_sim(X, X[:, first:first + 1].contiguous()))[:, 0].clamp_min_(0)  # squared distances (N,)
    pot = float(closest.sum(dtype=torch.float64))
    for _ in range(1, K):
        thresholds = torch.from_numpy(rng.uniform(size=n_trials) * pot).to(X.device)
        cum = torch.cumsum(closest, 0, dtype=torch.float64)
        cand = torch.searchsorted(cum, thresholds).clamp_(max=n - 1)
        D = (-ops.euc_sim(X, X[:, cand].contiguous())).clamp_min_(0)  # (N, n_trials)
        D = torch.minimum(D, closest[:, None])
        pots = D.sum(0, dtype=torch.float64)
        best = int(torch.argmin(pots))
        pot = float(pots[best])
        closest = D[:, best].contiguous()
        idx.append(int(cand[best]))
    return idx, X[:, torch.tensor(idx, device=X.device)].contiguous()


def sklearn_style_kmeans(C, K, *, random_state=0, n_init=10, max_iter=300, tol=1e-4):
    """``KMeans(n_clusters=K, random_state=random_state, init='k-means++', n_init=n_init).fit(C.T)`` as a recipe:
    -> (cluster centres (d,K) fp32 on C's device, inertia = mean squared distance, seed indices of every init)."""
    dev = ops.L.require_device(C)  # no CPU fallback: raises without a HIP device
    X = C.to(device=dev, dtype=torch.float32).contiguous()
    mean = X.mean(dim=1, keepdim=True)
    tol_ = float(X.var(dim=1, unbiased=False).mean()) * tol  # sklearn's _tolerance
    X = (X - mean).contiguous()
    rng = np.random.RandomState(random_state)
    best = None
    seeds = []
    for _ in range(n_init):
        idx, c0 = greedy_kmeanspp(X, K, rng)
        seeds.append(idx)
        res = ops.kmeans_fit(X, c0, max_iter, tol_)
        inertia = float(res["inertia"])
        if np.isfinite(inertia) and (best is None or inertia < best[0]):
            best = (inertia, res["centroids"])
    if best is None:
        raise RuntimeError("every k-means initialisation produced an empty cluster")
    return (best[1] + mean).contiguous(), best[0], seeds

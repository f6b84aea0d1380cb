"""ETAnchor -- the reference's EigenTrajectory/anchor.py interface on HIP kernels.

Reference: EigenTrajectory/anchor.py:5-88.  Same parameter (``C_anchor`` (k,S)).
``anchor_generation`` clusters the projected coefficients with this build's HIP
BatchKMeans (farthest-first seeding + Lloyd with exact sums; deterministic for a
given input) where the reference calls ``sklearn.cluster.KMeans(n_init=10)``
(anchor.py:65-71), which is not bit-reproducible even by itself with more than one
thread.  Anchors therefore agree in quality (inertia), not in value;
parity of the model is defined on loaded checkpoints (same anchors in -> same
trajectories out).
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

    def anchor_generation(self, pred_traj_norm, U_pred_trunc, *, n_redo=1, max_iter=100, tol=1e-4, seed=0):
        r"""Anchor generation on EigenTrajectory space (anchor.py:54-74; call once before training)."""
        C_pred = self.to_ET_space(pred_traj_norm, evec=U_pred_trunc)  # (k,N)
        self.generate_from_coefficients(C_pred, n_redo=n_redo, max_iter=max_iter, tol=tol, seed=seed)

    def generate_from_coefficients(self, C_pred, *, n_redo=1, max_iter=100, tol=1e-4, seed=0):
        r"""Cluster ET coefficients (k,N) into S anchors with the HIP BatchKMeans."""
        n = C_pred.shape[1]
        if n < self.s:
            raise ValueError(f"anchor generation needs at least num_samples={self.s} trajectories, got {n}")
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

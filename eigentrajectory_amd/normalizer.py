"""TrajNorm -- same interface as the reference's EigenTrajectory/normalizer.py, HIP-backed.

Reference: EigenTrajectory/normalizer.py:4-62.  The object is stateful like the
reference's (``calculate_params`` caches what ``normalize`` / ``denormalize`` use,
normalizer.py:15), but the cache it keeps is the compact ``nrm`` (4,N) block the
fused kernels exchange; ``traj_ori`` / ``traj_rot`` / ``traj_sca`` are materialised
on demand (model.py:86-87 reads ``traj_ori``; scripts call ``get_params``).
"""
from __future__ import annotations

import torch

from . import ops


class TrajNorm:
    r"""Normalize trajectory with shape (num_peds, length_of_time, 2)

    Args:
        ori (bool): Whether to normalize the trajectory with the origin
        rot (bool): Whether to normalize the trajectory with the rotation
        sca (bool): Whether to normalize the trajectory with the scale
    """

    def __init__(self, ori=True, rot=True, sca=True):
        self.ori, self.rot, self.sca = ori, rot, sca
        self._ori = self._rot = self._sca = None
        self._nrm = None      # (4,N): ox, oy, dx, dy (written by the projection kernel)
        self._t_obs = None

    # -- state -------------------------------------------------------------------------------
    def _from_nrm(self, nrm, t_obs):
        """Adopt the state cached by a fused projection (no kernel launch)."""
        self._nrm, self._t_obs = nrm, t_obs
        self._ori = self._rot = self._sca = None

    def _materialise(self):
        if self._nrm is not None and self._ori is None and self._rot is None and self._sca is None:
            self._ori, self._rot, self._sca = ops.norm_params_from_nrm(self._nrm, self.ori, self.rot, self.sca)

    def calculate_params(self, traj):
        r"""Calculate the normalization parameters (normalizer.py:17-29)"""
        self._nrm = None
        self._t_obs = traj.shape[1]
        self._ori, self._rot, self._sca = ops.norm_params(traj, self.ori, self.rot, self.sca)

    @property
    def traj_ori(self):
        self._materialise()
        return self._ori

    @property
    def traj_rot(self):
        self._materialise()
        return self._rot

    @property
    def traj_sca(self):
        self._materialise()
        return self._sca

    def get_params(self):
        r"""Get the normalization parameters"""
        return self.ori, self.rot, self.sca, self.traj_ori, self.traj_rot, self.traj_sca

    def set_params(self, ori, rot, sca, traj_ori, traj_rot, traj_sca):
        r"""Set the normalization parameters"""
        self.ori, self.rot, self.sca = ori, rot, sca
        self._nrm = None
        self._ori, self._rot, self._sca = traj_ori, traj_rot, traj_sca

    # -- transforms --------------------------------------------------------------------------
    def normalize(self, traj):
        r"""Normalize the trajectory (normalizer.py:42-51)"""
        return ops.normalize(traj, self.traj_ori if self.ori else None, self.traj_rot if self.rot else None,
                             self.traj_sca if self.sca else None).to(traj.device)

    def denormalize(self, traj):
        r"""Denormalize the trajectory (normalizer.py:53-62)"""
        return ops.denormalize(traj, self.traj_ori if self.ori else None, self.traj_rot if self.rot else None,
                               self.traj_sca if self.sca else None).to(traj.device)

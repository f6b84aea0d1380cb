"""Small host-side helpers the reference's users rely on (utils/utils.py, utils/metrics.py).

Only what the hot-path boundary needs: the attribute-style config dict the
wrapper is constructed with (utils/utils.py:32-39), the flip augmentation that
shapes the fit input (utils/utils.py:69-86) and ADE/FDE, the parity instrument
(utils/metrics.py:73-102).
"""
from __future__ import annotations

import torch


class DotDict(dict):
    r"""dot.notation access to dictionary attributes (utils/utils.py:32-39)"""

    __getattr__ = dict.get
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__
    __getstate__ = dict
    __setstate__ = dict.update


def default_hyper_params(**overrides):
    """The reference's config values for the descriptor path (config/eigentrajectory-{baseline}-eth.json:5-18)."""
    hp = DotDict(traj_dim=2, obs_len=8, pred_len=12, k=6, static_dist=0.419, num_samples=20, obs_svd=True,
                 pred_svd=True)
    hp.update(overrides)
    return hp


def augment_trajectory(obs_traj, pred_traj, flip=True, reverse=True):
    r"""Flip and reverse the trajectory (utils/utils.py:69-86; with the defaults only the y-flip is applied)."""
    if flip:
        sign = torch.tensor([[[1.0, -1.0]]], dtype=obs_traj.dtype, device=obs_traj.device)
        obs_traj = torch.cat([obs_traj, obs_traj * sign], dim=0)
        pred_traj = torch.cat([pred_traj, pred_traj * sign], dim=0)
    elif reverse:
        full_traj = torch.cat([obs_traj, pred_traj], dim=1)  # NTC
        obs_traj = torch.cat([obs_traj, full_traj.flip(1)[:, :obs_traj.size(1)]], dim=0)
        pred_traj = torch.cat([pred_traj, full_traj.flip(1)[:, obs_traj.size(1):]], dim=0)
    return obs_traj, pred_traj


def compute_batch_ade(pred, gt):
    r"""ADE per pedestrian (utils/metrics.py:73-86): pred (S,N,T,2), gt (N,T,2) -> (N,)"""
    temp = (pred - gt).norm(p=2, dim=-1)
    return temp.mean(dim=2).min(dim=0)[0]


def compute_batch_fde(pred, gt):
    r"""FDE per pedestrian (utils/metrics.py:89-102)"""
    temp = (pred - gt).norm(p=2, dim=-1)
    return temp[:, :, -1].min(dim=0)[0]

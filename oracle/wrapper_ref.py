"""numpy restatement of the reference wrapper on top of the C oracle.

TEST INFRASTRUCTURE ONLY (see oracle/et_oracle.c).  Restates
EigenTrajectory/model.py:58-125 (forward) and utils/metrics.py:73-102
(ADE/FDE) so that wrapper-level golden vectors (G6) can be checked without
the reference, and so the HIP wrapper can be compared against it on a GPU box.
"""
from __future__ import annotations

import numpy as np

from . import et_oracle as eo


def zero_stub(k, s):
    """baseline that predicts C_pred_refine = 0 (anchors only)."""
    return lambda x: np.zeros((k, x.shape[1], s), np.float32)


def linear_stub(w):
    """C_pred_refine[k',n,s] = sum_j W[s,k',j] * cat(C_obs, obs_ori)[j,n]  (tools/make_golden.py LinearStub)."""
    w = np.asarray(w, np.float32)
    return lambda x: np.einsum("skj,jn->kns", w, x).astype(np.float32)


def forward(params, obs, pred, predictor, static_dist):
    """model.py:58-125.  params: dict with the reference's state_dict keys.

    Returns dict(recon_traj (S,N,Tp,2), C_obs, obs_ori, and the three losses when pred is given).
    """
    Uom, Upm = params["ET_m_descriptor.U_obs_trunc"], params["ET_m_descriptor.U_pred_trunc"]
    Uos, Ups = params["ET_s_descriptor.U_obs_trunc"], params["ET_s_descriptor.U_pred_trunc"]
    Am, As = params["ET_m_anchor.C_anchor"], params["ET_s_anchor.C_anchor"]
    c_obs, c_pred_gt, nrm, _ = eo.norm_project(obs, pred, Uom, Upm, Uos, Ups, 2, static_dist)  # model.py:73-83
    obs_ori = nrm[:2].copy()
    obs_ori -= obs_ori.mean(axis=1, keepdims=True, dtype=np.float32)  # model.py:89
    x = np.concatenate([c_obs, obs_ori], axis=0)  # typical pre-hook, baseline/sgcn/bridge.py:6-7
    c_refine = np.ascontiguousarray(predictor(x), dtype=np.float32)  # (k,N,S) model.py:93-95
    recon = eo.anchor_reconstruct(c_refine, obs, Am, As, Upm, Ups, 2, static_dist)  # model.py:98-105
    out = dict(recon_traj=recon, C_obs=c_obs, obs_ori=obs_ori)
    if pred is not None:
        flag = eo.moving_flags(obs, static_dist)
        anchors = np.where(flag[None, :, None], Am[:, None, :], As[:, None, :])
        c_pred = anchors + c_refine  # model.py:110-111 (anchor.py:87)
        err_c = np.sqrt(((c_pred - c_pred_gt[:, :, None]) ** 2).sum(axis=0, dtype=np.float32))  # model.py:119
        err_d = np.sqrt(((recon - pred[None]) ** 2).sum(axis=-1, dtype=np.float32))  # model.py:120
        out["loss_eigentraj"] = err_c.min(axis=-1).mean(dtype=np.float32)
        out["loss_euclidean_ade"] = err_d.mean(axis=-1, dtype=np.float32).min(axis=0).mean(dtype=np.float32)
        out["loss_euclidean_fde"] = err_d[:, :, -1].min(axis=0).mean(dtype=np.float32)
    return out


def batch_ade(pred, gt):
    """utils/metrics.py:73-86"""
    temp = np.sqrt(((pred - gt[None]) ** 2).sum(axis=-1, dtype=np.float32))
    return temp.mean(axis=2, dtype=np.float32).min(axis=0)


def batch_fde(pred, gt):
    """utils/metrics.py:89-102"""
    temp = np.sqrt(((pred - gt[None]) ** 2).sum(axis=-1, dtype=np.float32))
    return temp[:, :, -1].min(axis=0)

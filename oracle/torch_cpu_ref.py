"""PyTorch-CPU restatement of the reference's op sequence for the descriptor hot path -- the CPU baseline.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/et_oracle.c): imported by tests/ and by ``bench.py``'s
``cpu_baseline`` leg, never by ``eigentrajectory_amd``.  Where ``et_oracle.c`` restates the ARITHMETIC in scalar
C (the parity checker), this file restates the ATen CALL SEQUENCE the reference executes on CPU tensors, so that
timing it on the GPU box's host cores (``torch.set_num_threads(os.cpu_count())``) is a fair stand-in for "the
reference's PyTorch-CPU path"  -- the reference itself cannot travel to the GPU box.  Written
from SURVEY.md §8(a); each function cites the lines whose op sequence it follows.  Pinned against the golden
fixtures in tests/test_torch_cpu_ref.py.
"""
from __future__ import annotations

import time

import torch


# ----------------------------------------------------------------------------------- TrajNorm
def norm_params(obs, use_sca):
    """normalizer.py:17-29: origin = last observed point, heading from the last two steps, scale 2/|d|."""
    ori = obs[:, [-1]]
    d = obs[:, -1] - obs[:, -3]
    th = torch.atan2(d[:, 1], d[:, 0])
    c, s = th.cos(), th.sin()
    rot = torch.stack([c, -s, s, c], dim=1).reshape(-1, 2, 2)  # [[cos, -sin], [sin, cos]] per pedestrian
    sca = (1.0 / d.norm(p=2, dim=-1)[:, None, None] * 2) if use_sca else None
    return ori, rot, sca


def normalize(traj, ori, rot, sca):
    """normalizer.py:42-51"""
    out = (traj - ori) @ rot
    return out * sca if sca is not None else out


def denormalize(traj, ori, rot, sca):
    """normalizer.py:53-62"""
    out = traj / sca if sca is not None else traj
    return out @ rot.transpose(-1, -2) + ori


# --------------------------------------------------------------------------------- descriptor
def truncated_svd(traj_norm, k):
    """descriptor.py:91-114: SVD of the (2T x N) matrix; only U[:, :k] is ever used (descriptor.py:134-135)."""
    n = traj_norm.size(0)
    M = traj_norm.reshape(n, -1).T
    U, S, Vt = torch.linalg.svd(M, full_matrices=False)
    return U[:, :k], S[:k], Vt[:k].T


def to_et_space(traj_norm, U):
    """descriptor.py:59-73: C = U^T M"""
    n = traj_norm.size(0)
    return U.T.detach() @ traj_norm.reshape(n, -1).T


def to_euclidean_space(C, U):
    """descriptor.py:75-89"""
    n = C.size(1)
    return (U.detach() @ C).T.reshape(n, -1, 2)


def reconstruction(C, U, ori, rot, sca):
    """descriptor.py:162-176: one to_Euclidean_space + denormalize per sample, stacked.  C (k,N,S)."""
    return torch.stack([denormalize(to_euclidean_space(C[:, :, s], U), ori, rot, sca) for s in range(C.size(2))], dim=0)


# ------------------------------------------------------------------------------------ k-means
def euc_sim(a, b):
    """kmeans.py:59-76: negative squared distance, (2 a^T b - |a|^2) - |b|^2 with in-place updates."""
    y = a.transpose(-2, -1) @ b
    y.mul_(2)
    y.sub_(a.pow(2).sum(dim=-2)[..., :, None])
    y.sub_(b.pow(2).sum(dim=-2)[..., None, :])
    return y


def farthest_first(data, K, first_index):
    """kmeans.py:78-112 for (d,N) data: every further centroid is the point least similar to the chosen ones."""
    d, n = data.shape
    cen = torch.zeros((d, K), dtype=data.dtype)
    cen[:, 0] = data[:, first_index]
    for i in range(1, K):
        sims = euc_sim(data, cen[:, :i].contiguous())
        cen[:, i] = data[:, sims.max(dim=-1)[0].argmin(dim=-1)]
    return cen


def compute_centroids(data, labels, K):
    """kmeans.py:160-182: one-hot masks, broadcast product (d,N,K), column sums / counts (empty cluster -> NaN)."""
    mask = torch.stack([labels == i for i in range(K)], dim=-1)
    return (data.unsqueeze(dim=-1) * mask.unsqueeze(dim=-3)).sum(dim=-2) / mask.sum(dim=-2, keepdim=True)


def lloyd(data, centroids, max_iter=100, tol=1e-4, deadline=None):
    """kmeans.py:228-240 -> dict(centroids, labels, n_iter, error, inertia, trace).  ``deadline`` (a
    time.perf_counter() value) bounds a timing run: the loop stops after the iteration during which it passed."""
    trace = []
    for j in range(max_iter):
        maxsims, labels = euc_sim(data, centroids).max(dim=-1)
        new = compute_centroids(data, labels, centroids.size(-1))
        error = (centroids - new).pow(2).sum()
        centroids = new
        inertia = (-maxsims).mean()
        trace.append((float(error), float(inertia)))
        if error <= tol or (deadline is not None and time.perf_counter() > deadline):
            break
    return dict(centroids=centroids, labels=labels, n_iter=j + 1, error=float(error), inertia=float(inertia), trace=trace)


# ----------------------------------------------------------------------------------- hot path
def hot_path(obs, pred, k=6, K=20, first_index=0, max_iter=100, tol=1e-4, stages=None, deadline=None):
    """One *step* of bench.py on CPU tensors, moving descriptor (norm_sca=True):
    fit (normalise + two SVDs) -> project (obs+pred) -> reconstruct (S=1) -> farthest-first + Lloyd on C_pred.
    ``stages`` (dict) receives the wall time of every stage.  -> dict(U_obs, U_pred, C_pred, recon, kmeans)."""
    t = time.perf_counter
    t0 = t()
    ori, rot, sca = norm_params(obs, True)
    obs_n, pred_n = normalize(obs, ori, rot, sca), normalize(pred, ori, rot, sca)
    U_obs = truncated_svd(obs_n, k)[0]
    U_pred = truncated_svd(pred_n, k)[0]
    t1 = t()
    ori, rot, sca = norm_params(obs, True)  # projection() normalises again (descriptor.py:157)
    obs_n, pred_n = normalize(obs, ori, rot, sca), normalize(pred, ori, rot, sca)
    C_obs, C_pred = to_et_space(obs_n, U_obs), to_et_space(pred_n, U_pred)
    t2 = t()
    recon = reconstruction(C_pred.unsqueeze(-1), U_pred, ori, rot, sca)
    t3 = t()
    x = C_pred.contiguous()
    c0 = farthest_first(x, K, first_index)
    t4 = t()
    km = lloyd(x, c0, max_iter, tol, deadline)
    t5 = t()
    if stages is not None:
        stages.update(fit=t1 - t0, project=t2 - t1, reconstruct=t3 - t2, kmeans_init=t4 - t3, kmeans_lloyd=t5 - t4,
                      total=t5 - t0, lloyd_iterations=km["n_iter"])
    return dict(U_obs=U_obs, U_pred=U_pred, C_obs=C_obs, C_pred=C_pred, recon=recon, c0=c0, kmeans=km)

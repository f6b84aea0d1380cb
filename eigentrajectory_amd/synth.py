"""Deterministic synthetic trajectory generators.

The reference ships no synthetic generator; its only inputs are the ETH/UCY
text files (utils/dataloader.py:154-241).  The benchmark and the parity
tests need seeded inputs of arbitrary N, so this module defines them once:

* ``synthetic_trajectories_np``  -- numpy PCG64 stream, bit-stable across
  machines; used by tests / golden fixtures (small N).
* ``synthetic_trajectories_torch`` -- same distribution drawn on a torch
  device generator; used by bench.py for N up to 1e7+ directly in HBM.

Model (per trajectory, 20 steps of 0.4 s like ETH/UCY):
  v0 ~ N((0.5, 0), 0.4^2 I) m/step, a_t ~ N(0, 0.05^2 I),
  vel_t = v0 + cumsum(a), p0 ~ N(0, 5^2 I), pos = p0 + cumsum(vel).
obs = pos[:, :obs_len], pred = pos[:, obs_len:], fp32 contiguous (N,T,2).
"""
from __future__ import annotations

import numpy as np


def synthetic_trajectories_np(n: int, seed: int = 0, obs_len: int = 8, pred_len: int = 12,
                              min_disp: float = 0.0):
    """Return (obs (n,obs_len,2), pred (n,pred_len,2)) float32 numpy arrays.

    ``min_disp`` > 0 re-draws rows whose ||obs[-1]-obs[-3]|| is below it
    (single-descriptor / norm_sca=True kernel benchmarks need finite scale;
    the reference avoids the degenerate rows by routing them to the
    norm_sca=False descriptor, EigenTrajectory/model.py:46-52).
    """
    rng = np.random.default_rng(seed)
    t = obs_len + pred_len

    def draw(m):
        v0 = rng.standard_normal((m, 1, 2)) * 0.4 + np.array([0.5, 0.0])
        acc = rng.standard_normal((m, t, 2)) * 0.05
        vel = v0 + np.cumsum(acc, axis=1)
        p0 = rng.standard_normal((m, 1, 2)) * 5.0
        return (p0 + np.cumsum(vel, axis=1)).astype(np.float32)

    pos = draw(n)
    if min_disp > 0.0:
        for _ in range(64):
            d = pos[:, obs_len - 1] - pos[:, obs_len - 3]
            bad = np.nonzero(np.hypot(d[:, 0], d[:, 1]) < min_disp)[0]
            if bad.size == 0:
                break
            pos[bad] = draw(bad.size)
    obs = np.ascontiguousarray(pos[:, :obs_len])
    pred = np.ascontiguousarray(pos[:, obs_len:])
    return obs, pred


def synthetic_trajectories_torch(n: int, device, seed: int = 0, obs_len: int = 8, pred_len: int = 12,
                                 min_disp: float = 0.0):
    """Same distribution as :func:`synthetic_trajectories_np`, generated on ``device``.

    Not bit-identical to the numpy stream (different generator); every
    consumer that needs CPU/GPU agreement copies these tensors instead of
    re-drawing them.
    """
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = obs_len + pred_len
    chunk = 1 << 21  # bound temporaries: 2M rows * 20 * 2 * 4 B = 335 MB
    obs = torch.empty((n, obs_len, 2), dtype=torch.float32, device=device)
    pred = torch.empty((n, pred_len, 2), dtype=torch.float32, device=device)
    mean = torch.tensor([0.5, 0.0], device=device)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        v0 = torch.randn((m, 1, 2), generator=g, device=device) * 0.4 + mean
        acc = torch.randn((m, t, 2), generator=g, device=device) * 0.05
        vel = v0 + torch.cumsum(acc, dim=1)
        p0 = torch.randn((m, 1, 2), generator=g, device=device) * 5.0
        pos = p0 + torch.cumsum(vel, dim=1)
        if min_disp > 0.0:
            d = pos[:, obs_len - 1] - pos[:, obs_len - 3]
            bad = d.norm(dim=-1) < min_disp
            # deterministic repair: push the last three observed steps along +x
            fix = torch.zeros_like(pos)
            fix[:, obs_len - 2:, 0] = min_disp
            fix[:, obs_len - 1:, 0] = 2.0 * min_disp
            pos = torch.where(bad[:, None, None], pos + fix, pos)
        obs[lo:lo + m] = pos[:, :obs_len]
        pred[lo:lo + m] = pos[:, obs_len:]
    return obs, pred


def gaussian_points_np(d: int, n: int, seed: int = 0, n_blobs: int = 0):
    """(d, n) float32 d-major point cloud for the k-means tests.

    ``n_blobs`` = 0 -> isotropic N(0, I) (what SURVEY.md §3.4 timed);
    otherwise a mixture of ``n_blobs`` unit blobs with centres ~ N(0, 4^2 I).
    """
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((d, n))
    if n_blobs > 0:
        centres = rng.standard_normal((d, n_blobs)) * 4.0
        which = rng.integers(0, n_blobs, size=n)
        x = x + centres[:, which]
    return np.ascontiguousarray(x.astype(np.float32))

"""CPU estimate (torch, fp64 distances) of what carried bounds would save in the Lloyd loop of bench.py's workload:
Hamerly's test (upper bound to the own centroid vs ONE lower bound to the rest) and a G-group variant (one lower
bound per group of centroids, groups by centroid norm).  Prints, per iteration, the share of points whose bounds
fail (they would need the full scan) and the share of 64-point groups with at least one such point.

    python tools/sim_bound_skipping.py [N=200000] [G=2]

Result on the synthetic trajectories (N = 2e5, K = 20, farthest-first): the data are heavy tailed (max |x| 2217, rms 29),
the outlying centroids keep moving by 1e-1 .. 1e+1 per iteration while the median gap of a point is ~33 in squared
units; 50-100 % of the points fail until iteration ~55, 10-25 % at iterations 90-100, and no 64-point group is ever
clean.  Nothing to gain for this workload within max_iter = 100, see DESIGN.md section 3.
"""
import sys
import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from eigentrajectory_amd.synth import synthetic_trajectories_np


def coefficients(n):
    """moving-descriptor C_pred (6, n) of the synthetic workload, plain torch on the CPU"""
    obs, pred = (torch.from_numpy(a).double() for a in synthetic_trajectories_np(n, seed=0, min_disp=1e-3))
    ori = obs[:, -1]
    d = obs[:, -1] - obs[:, -3]
    th = torch.atan2(d[:, 1], d[:, 0])
    rot = torch.stack([torch.stack([th.cos(), -th.sin()], 1), torch.stack([th.sin(), th.cos()], 1)], 1)
    sca = 1.0 / (obs[:, -1] - obs[:, -3]).norm(dim=-1) * 2
    pn = ((pred - ori[:, None]) @ rot) * sca[:, None, None]
    M = pn.reshape(n, -1).T
    U = torch.linalg.svd(M, full_matrices=False)[0][:, :6]
    return (U.T @ M).contiguous()


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    K = 20
    X = coefficients(n)
    cen = torch.zeros((6, K), dtype=torch.double)
    cen[:, 0] = X[:, 0]
    for i in range(1, K):  # farthest-first
        cen[:, i] = X[:, torch.cdist(X.T, cen[:, :i].T).min(1).values.argmax()]
    print(f"N {n}  max|x| {float(X.norm(dim=0).max()):.1f}  rms {float(X.norm(dim=0).pow(2).mean().sqrt()):.1f}")
    rows = torch.arange(n)
    for it in range(100):
        D = torch.cdist(X.T, cen.T)
        labels = D.argmin(1)
        new = torch.stack([X[:, labels == j].mean(1) for j in range(K)], 1)
        e = (new - cen).norm(dim=0)
        if it == 0:
            order = new.norm(dim=0).argsort()
            grp = torch.empty(K, dtype=torch.long)
            for g, idx in enumerate(np.array_split(np.arange(K), G)):
                grp[order[idx]] = g
            fail = torch.ones(n, dtype=torch.bool)
            fail1 = fail.clone()
            u, u1, l1 = (torch.zeros(n, dtype=torch.double) for _ in range(3))
            lg = torch.zeros((n, G), dtype=torch.double)
            changed = n
        else:
            fail, fail1 = u >= lg.min(1).values, u1 >= l1
            ch = labels != prev
            changed = int(ch.sum())
            assert not bool((ch & ~fail).any()) and not bool((ch & ~fail1).any())  # the tests are sound
        Dm = D.clone()
        Dm[rows, labels] = float("inf")
        da = D[rows, labels]
        u[fail], u1[fail1] = da[fail], da[fail1]
        l1[fail1] = Dm.min(1).values[fail1]
        for g in range(G):
            lg[fail, g] = Dm[:, grp == g].min(1).values[fail]
            lg[:, g] -= e[grp == g].max()
        u += e[labels]
        u1 += e[labels]
        l1 -= e.max()
        g64 = fail[: n // 64 * 64].view(-1, 64).any(1).double().mean()
        print(f"it {it:3d}  max move {float(e.max()):.2e}  changed {changed:7d}  hamerly fails {float(fail1.double().mean()):.4f}"
              f"  {G}-group fails {float(fail.double().mean()):.4f}  64-point groups touched {float(g64):.3f}")
        prev, cen = labels, new


if __name__ == "__main__":
    main()

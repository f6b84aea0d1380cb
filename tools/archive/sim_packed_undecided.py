"""Which fraction of the points would the packed-copy certification of csrc/et_kmeans.hip (packed_assign_body) leave undecided,
per Lloyd iteration, on the data bench.py clusters?  A CPU estimate in fp64 with the error terms of the kernel: the current
fp32 filter, f16 coordinates about the origin, f16 coordinates about the mean (adopted), f16 for point AND centroid.
Run here (no GPU): python tools/sim_packed_undecided.py [N]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd.synth import synthetic_trajectories_np
from oracle import et_oracle as O
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400000
obs, pred = synthetic_trajectories_np(N, seed=0, min_disp=1e-3)
# simple: normalise (sca on), SVD via numpy
on, pn = O.normalize(obs, obs, True), O.normalize(obs, pred, True)
M = pn.reshape(N, 24).astype(np.float64)
w, V = np.linalg.eigh(M.T @ M)
U = V[:, ::-1][:, :6].astype(np.float32)
X = np.ascontiguousarray((pn.reshape(N, 24) @ U).T.astype(np.float32))  # (6,N)
print("coef std", X.std(axis=1), "mean", X.mean(axis=1), "max", np.abs(X).max())
K = 20
np.random.seed(0)
first = np.random.randint(N)
C = O.kmeans_init_farthest(X, K, first)
C = C[0] if isinstance(C, tuple) else C
Xd = X.astype(np.float64)
mu = Xd.mean(axis=1, keepdims=True)
labels = None
for it in range(40):
    Cd = C.astype(np.float64)
    Y = 2 * Xd.T @ Cd - (Cd ** 2).sum(0)[None, :]  # (N,K) = G_j
    order = np.argsort(-Y, axis=1)[:, :2]
    best = Y[np.arange(N), order[:, 0]]; sec = Y[np.arange(N), order[:, 1]]
    newl = order[:, 0]
    if labels is not None:
        changed = (newl != labels).mean()
        r = np.sqrt((Xd ** 2).sum(0)); Cn = np.sqrt((Cd ** 2).sum(0))
        rt = np.sqrt(((Xd - mu) ** 2).sum(0)); Ct = np.sqrt(((Cd - mu) ** 2).sum(0))
        # margin of the OLD label vs best other
        Yl = Y[np.arange(N), labels]
        other = np.where(order[:, 0] == labels, sec, best)
        margin = Yl - other
        jo = np.where(order[:, 0] == labels, order[:, 1], order[:, 0])
        def und(eps_l, eps_o):
            return (margin <= eps_l + eps_o).mean()
        # current scheme: eps_j = 2^-16 (r+C)^2 + 2^-20 (r + C) (scaled; unscale: ^2 terms scale-free)
        e_cur = lambda rr, cc: 2.0 ** -16 * (rr + cc) ** 2
        cur = und(e_cur(r, Cn[labels]) * 0.5, e_cur(r, Cn[jo]))
        # hi-only uncentred: + 2^-10 r C each side
        e_h = lambda rr, cc: 2.0 ** -10 * rr * cc + 2.0 ** -16 * (rr + cc) ** 2
        hu = und(e_h(r, Cn[labels]), e_h(r, Cn[jo]))
        hc = und(e_h(rt, Ct[labels]) + 2.0**-20*(r+Cn[labels])**2, e_h(rt, Ct[jo]) + 2.0**-20*(r+Cn[jo])**2)
        # bf16-like 8-bit? skip.  f16 hi for x AND c (no lo): 2^-9 r C
        e_hh = lambda rr, cc: 2.0 ** -9 * rr * cc + 2.0 ** -16 * (rr + cc) ** 2
        hhc = und(e_hh(rt, Ct[labels]) + 2.0**-20*(r+Cn[labels])**2, e_hh(rt, Ct[jo]) + 2.0**-20*(r+Cn[jo])**2)
        print(f"it {it:3d} changed {changed*100:6.3f}%  undecided: current {cur*100:6.3f}%  hi-only {hu*100:6.3f}%  hi-only centred {hc*100:6.3f}%  hi/hi centred {hhc*100:6.3f}%")
    labels = newl
    # update
    Cn_ = np.zeros_like(Cd)
    for j in range(K):
        m = labels == j
        Cn_[:, j] = Xd[:, m].mean(axis=1) if m.any() else np.nan
    C = Cn_.astype(np.float32)
print("median margin", np.median(margin), "1% quantile of |margin|", np.quantile(np.abs(margin), [0.001, 0.01, 0.05]))
print("eps centred hi-only: median", np.median(e_h(rt, Ct[labels])), "max", e_h(rt, Ct[labels]).max(), " uncentred median", np.median(e_h(r, Cn[labels])))
print("rt median", np.median(rt), "Ct", Ct, "r median", np.median(r))
